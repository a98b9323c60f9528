// The reference's model-load and frame flow through the C++ host mirror (include/sdfbox.hpp):
//   Logic.MakeData(args[0]) -> Program.Load -> Logic.Heading/Position -> Program.Draw
// (Program.cs:38-77).  Built and run by tests/test_cpp_host.py.
//   cpp_host <model path, with or without extension> <W> <H> <out.raw> [<out_display.raw>]
// exit 0 ok, 3 = no usable GPU, anything else = failure.
#include "sdfbox.hpp"
#include <cstdio>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage\n"); return 2; }
    using namespace SDFbox;
    try {
        Logic logic;
        // file-type dispatch, as the reference does it
        if (Logic::FormatOf("a/b.asdf") != FileFormat::ASDF || Logic::FormatOf("x.ply") != FileFormat::Stanford ||
            Logic::FormatOf("x.obj") != FileFormat::Wavefront || Logic::FormatOf("x.stl") != FileFormat::Invalid) return 10;
        if (!Logic::AutocompleteFile("/nonexistent/model").empty()) return 11;
        OctData model;
        try {
            model = logic.MakeData(argv[1]);
        } catch (const Error &e) {
            if (e.code == SDFHIP_ERR_DEVICE) { fprintf(stderr, "%s\n", e.what()); return 3; }   // mesh import needs the GPU
            throw;
        }
        printf("Length %d buffer_size %u\n", model.Length(), logic.State.buffer_size);
        int W = atoi(argv[2]), H = atoi(argv[3]);
        logic.Resize(W, H);
        logic.SetHeading(0.0f, 0.0f);             // Program.cs:54-55
        logic.SetPosition(0.5f, 0.5f, 0.1f);
        Program program;
        try {
            program.Load(model);
        } catch (const Error &e) {
            if (e.code == SDFHIP_ERR_DEVICE) { fprintf(stderr, "%s\n", e.what()); return 3; }
            throw;
        }
        std::vector<float> frame;
        program.Draw(logic.State, W, H, frame);
        FILE *f = fopen(argv[4], "wb");
        if (!f || fwrite(frame.data(), 4, frame.size(), f) != frame.size()) return 12;
        fclose(f);
        if (argc > 5) {
            std::vector<uint8_t> disp;
            program.DrawDisplay(logic.State, W, H, false, disp);
            f = fopen(argv[5], "wb");
            if (!f || fwrite(disp.data(), 1, disp.size(), f) != disp.size()) return 13;
            fclose(f);
        }
        {   // Logic.Update / MouseMove through the mirror: a step forward and back again returns the camera
            Logic walk;
            walk.Update(2.0f, SDFHIP_KEY_FORWARD);
            if (walk.State.position[2] <= 0.1f) return 16;
            walk.Update(2.0f, SDFHIP_KEY_BACK);
            if (walk.State.position[2] < 0.0999f || walk.State.position[2] > 0.1001f) return 17;
            walk.MouseMove(128.0f, 0.0f);
            if (walk.HeadingY() != 1.0f) return 18;
        }
        program.Load(model);                       // reload swaps the scene (Program.cs:59-65)
        program.Draw(logic.State, W, H, frame);
        {   // the upload's choices as arguments: no lookup grid at all (the cursor-stack kernel walks the tree) gives the same frame
            sdfhip_upload_options opt;
            sdfhip_upload_options_default(&opt);
            opt.top_grid_level = 0;
            program.Load(model, opt);
            std::vector<float> plain;
            program.Draw(logic.State, W, H, plain);
            if (plain.size() != frame.size() || memcmp(plain.data(), frame.data(), frame.size() * 4) != 0) return 20;
            program.Load(model);
        }
        {   // the same frame over three ranks (the one GPU named three times): ProgramMulti.Draw is still one call
            ProgramMulti multi({0, 0, 0});
            multi.Load(model);
            std::vector<float> frame3;
            multi.Draw(logic.State, W, H, frame3);
            if (frame3.size() != frame.size() || memcmp(frame3.data(), frame.data(), frame.size() * 4) != 0) return 19;
            multi.Draw(logic.State, W, H, frame3);
            if (memcmp(frame3.data(), frame.data(), frame.size() * 4) != 0) return 19;
        }
        try {                                      // errors are exceptions of one type, never a crash
            Logic other;
            other.MakeData("/nonexistent/model");
            return 14;
        } catch (const Error &e) {
            if (e.code != SDFHIP_ERR_IO) return 15;
        }
        return 0;
    } catch (const SDFbox::Error &e) {
        fprintf(stderr, "Error %d: %s\n", e.code, e.what());
        return 20;
    }
}
