// sdfbox.hpp -- the host side of SdfBox's hot path, in C++, above the C ABI of sdfhip.h.
//
// The reference's host is C# (no .NET toolchain in the build image), so this header mirrors
// the classes a SdfBox maintainer knows -- same names, same argument meaning, same flow --
// as a header-only C++ layer; it is what tests/cpp_host.cpp drives.  Errors surface as
// SDFbox::Error (the reference lets native exceptions cross the FFI; here they are
// status codes below this layer and one C++ exception type above it).
//
//   SDFbox::OctLean, SDFbox::OctData      SdfBox/Program.cs:339-350, 503-667
//   SDFbox::OctData::NativeOctData        SdfBox/Program.cs:579-667 (Generate: .asdf, or
//                                         .ply/.obj -> SdfGen -> Save next to the mesh)
//   SDFbox::FileFormat, Logic::FormatOf, Logic::AutocompleteFile, Logic::MakeData
//                                         SdfBox/Logic.cs:87-151, 399-404
//   SDFbox::Logic::State/Heading/Position SdfBox/Logic.cs:30-78
//   SDFbox::Model::MaxDepth               SdfBox/Model.cs:18
//   SDFbox::Program::Load/Draw            SdfBox/Program.cs:79-110, 167-173 (compute pass only)
#pragma once
#include "sdfhip.h"

#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace SDFbox {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
inline void Check(int rc)
{
    if (rc != SDFHIP_OK) throw Error(rc, std::string("sdfhip: ") + sdfhip_last_error());
}

enum class FileFormat { Invalid = -1, ASDF = 0, Stanford = 1, Wavefront = 2 };   // Logic.cs:399-404

struct OctLean {                     // Program.cs:339-350
    int32_t Parent, Children;
};
typedef sdfhip_info Info;            // Logic.cs:407-420, byte for byte

struct Model {
    static constexpr int MaxDepth = 10;   // Model.cs:18
};

class OctData {                      // Program.cs:503-577 (without the texture swizzle: not needed)
public:
    std::vector<OctLean> Structs;
    std::vector<uint8_t> Values;     // 8 per node, corner k = x + 2y + 4z
    int Length() const { return (int)Structs.size(); }

    // Program.cs:579-667
    struct NativeOctData {
        // Generate(path, type): load an .asdf, or import a mesh, build the ASDF on the GPU and
        // cache it as <mesh>.asdf next to the source (Program.cs:613-650)
        static OctData Generate(const std::string &path, FileFormat type, int device = 0)
        {
            sdfhip_octdata nod{};
            if (type == FileFormat::ASDF) {
                Check(sdfhip_asdf_load(path.c_str(), &nod));
            } else {
                sdfhip_points vertices{};
                if (type == FileFormat::Stanford) Check(sdfhip_load_ply(path.c_str(), &vertices));
                else if (type == FileFormat::Wavefront) Check(sdfhip_load_obj(path.c_str(), &vertices));
                else throw Error(SDFHIP_ERR_ARG, "NotImplemented: unknown file format");
                int rc = sdfhip_sdfgen(device, vertices.data, vertices.count, Model::MaxDepth, &nod, nullptr);
                sdfhip_points_free(&vertices);
                Check(rc);
                std::string saveto = path.substr(0, path.find_last_of('.')) + ".asdf";   // Path.ChangeExtension
                Check(sdfhip_asdf_save(&nod, saveto.c_str()));
            }
            OctData d = FromNative(nod);
            sdfhip_octdata_free(&nod);                                                  // NativeOctData.Free
            return d;
        }
    };

    void Save(const std::string &path) const
    {
        sdfhip_octdata raw{ (uint32_t)Length(), (int32_t *)Structs.data(), (uint8_t *)Values.data() };
        Check(sdfhip_asdf_save(&raw, path.c_str()));
    }

private:
    static OctData FromNative(const sdfhip_octdata &raw)            // ManagedStructs / ManagedValues
    {
        OctData d;
        d.Structs.resize(raw.length);
        d.Values.resize((size_t)raw.length * 8);
        memcpy(d.Structs.data(), raw.structs, (size_t)raw.length * 8);
        memcpy(d.Values.data(), raw.values, (size_t)raw.length * 8);
        return d;
    }
};

class Logic {                        // Logic.cs:15-355, the camera/model part
public:
    static constexpr int xSize = 720, ySize = 720;                   // Logic.cs:17-18
    Info State;

    Logic() { sdfhip_info_default(&State, (float)xSize, (float)ySize); }
    // Logic.Heading (X = pitch, Y = yaw), Logic.cs:46-55
    void SetHeading(float x, float y) { headingX = x; headingY = y; sdfhip_info_set_heading(&State, x, y); }
    float HeadingX() const { return headingX; }
    float HeadingY() const { return headingY; }
    // Logic.Position, Logic.cs:60-78 (also refreshes State.limit)
    void SetPosition(float x, float y, float z) { sdfhip_info_set_position(&State, x, y, z); }
    void Resize(int width, int height) { State.screen_size[0] = (float)width; State.screen_size[1] = (float)height; }
    // the camera part of Logic.Update (Logic.cs:239-272): keys is a mask of SDFHIP_KEY_*
    void Update(float seconds, uint32_t keys)
    {
        float h[2] = { headingX, headingY };
        sdfhip_camera_update(&State, h, mSpeed, keys, seconds);
        headingX = h[0]; headingY = h[1];
    }
    void MouseMove(float dx, float dy)                                // Logic.cs:290-293
    {
        float h[2] = { headingX, headingY };
        sdfhip_camera_mouse_move(&State, h, dx, dy);
        headingX = h[0]; headingY = h[1];
    }
    void MouseWheel(float delta) { mSpeed = sdfhip_camera_mouse_wheel(mSpeed, delta); }   // Logic.cs:202-205
    float mSpeed = 0.5f;                                              // Logic.cs:28

    static FileFormat FormatOf(const std::string &filename)          // Logic.cs:126-138
    {
        auto ends = [&](const char *e) { size_t n = strlen(e); return filename.size() >= n && filename.compare(filename.size() - n, n, e) == 0; };
        if (ends(".asdf")) return FileFormat::ASDF;
        if (ends(".ply")) return FileFormat::Stanford;
        if (ends(".obj")) return FileFormat::Wavefront;
        return FileFormat::Invalid;
    }
    static std::string AutocompleteFile(const std::string &filename)  // Logic.cs:139-151 ("" for null)
    {
        auto exists = [](const std::string &p) { return std::ifstream(p).good(); };
        for (const char *ext : { "", ".asdf", ".ply", ".obj" })
            if (exists(filename + ext)) return filename + ext;
        return std::string();
    }
    // Logic.MakeData, Logic.cs:87-105
    OctData MakeData(const std::string &name, int device = 0)
    {
        std::string filename = AutocompleteFile(name);
        if (filename.empty()) throw Error(SDFHIP_ERR_IO, "MakeData: no such model: " + name);
        OctData data = OctData::NativeOctData::Generate(filename, FormatOf(filename), device);
        State.buffer_size = (uint32_t)data.Length();
        return data;
    }

private:
    float headingX = 0, headingY = 0;
};

class Program {                      // Program.cs: the compute pass of Draw, and model (re)load
public:
    explicit Program(int device = 0) : device(device) {}
    ~Program() { if (scene) sdfhip_scene_free(scene); }
    Program(const Program &) = delete;
    Program &operator=(const Program &) = delete;

    // CreateResources' scene bindings / the reload swap of Program.cs:59-65: upload the new
    // model, then drop the old one
    void Load(const OctData &model)
    {
        sdfhip_scene *fresh = nullptr;
        Check(sdfhip_scene_upload(device, &model.Structs[0].Parent, model.Values.data(), (uint32_t)model.Length(), &fresh));
        if (scene) sdfhip_scene_free(scene);
        scene = fresh;
    }
    // ... with the upload's choices taken by the host (sdfhip_upload_options: start from sdfhip_upload_options_default, -1 = choose):
    // e.g. top_grid_level = 0 for no lookup grid at all on a device short of memory
    void Load(const OctData &model, const sdfhip_upload_options &options)
    {
        sdfhip_scene *fresh = nullptr;
        Check(sdfhip_scene_upload_ex(device, &model.Structs[0].Parent, model.Values.data(), (uint32_t)model.Length(), &options, &fresh));
        if (scene) sdfhip_scene_free(scene);
        scene = fresh;
    }
    // Draw's UpdateBuffer(info) + DispatchSized(W, H, 1), Program.cs:81,94 -> RGBA32F frame
    void Draw(const Info &state, int width, int height, std::vector<float> &frame, uint32_t flags = 0)
    {
        if (!scene) throw Error(SDFHIP_ERR_ARG, "Draw: no model loaded");
        frame.resize((size_t)width * height * 4);
        Check(sdfhip_render(scene, &state, (uint32_t)width, (uint32_t)height, flags, frame.data(), nullptr));
    }
    // Draw including the display pass (DisplayFrag.hlsl via Program.cs:96-99) -> RGBA8 frame
    void DrawDisplay(const Info &state, int width, int height, bool debugGizmos, std::vector<uint8_t> &frame, uint32_t flags = 0)
    {
        if (!scene) throw Error(SDFHIP_ERR_ARG, "Draw: no model loaded");
        frame.resize((size_t)width * height * 4);
        Check(sdfhip_render_display(scene, &state, (uint32_t)width, (uint32_t)height, flags, debugGizmos ? 1 : 0, frame.data(), nullptr));
    }

private:
    int device;
    sdfhip_scene *scene = nullptr;
};

// The same two calls with the frame rendered by several GPUs of the node (sdfhip_multi_*: the scene on every device, the frame's
// bands dealt to them, sparse shares gathered into devices[0], assembled there): Program.Draw stays ONE call.
class ProgramMulti {
public:
    explicit ProgramMulti(std::vector<int> devices) : devices(std::move(devices)) {}
    ~ProgramMulti() { if (multi) sdfhip_multi_free(multi); }
    ProgramMulti(const ProgramMulti &) = delete;
    ProgramMulti &operator=(const ProgramMulti &) = delete;

    void Load(const OctData &model)
    {
        sdfhip_multi *fresh = nullptr;
        Check(sdfhip_multi_create(devices.data(), (uint32_t)devices.size(), &model.Structs[0].Parent, model.Values.data(), (uint32_t)model.Length(), &fresh));
        if (multi) sdfhip_multi_free(multi);
        multi = fresh;
    }
    void Draw(const Info &state, int width, int height, std::vector<float> &frame, uint32_t flags = 0)
    {
        if (!multi) throw Error(SDFHIP_ERR_ARG, "Draw: no model loaded");
        frame.resize((size_t)width * height * 4);
        Check(sdfhip_multi_render(multi, &state, (uint32_t)width, (uint32_t)height, flags, frame.data(), nullptr));
    }

private:
    std::vector<int> devices;
    sdfhip_multi *multi = nullptr;
};

}  // namespace SDFbox
