// Dev microbenchmark: does a HIP graph shorten a frame of the multi-device pipeline?
// The pipeline's shape on one device: N "rank" streams each run one small kernel whose ARGUMENT changes every frame (the camera
// block travels in the kernel arguments) and record an event; the "rank 0" stream waits for the N events and runs one kernel; the
// host waits for that.  Timed per frame, host clock, a frame at a time (submit + wait):
//   direct : hipLaunchKernelGGL / hipEventRecord / hipStreamWaitEvent per frame
//   graph  : the same captured once as a graph; per frame hipGraphExecKernelNodeSetParams for the N rank kernels + hipGraphLaunch
//   graph0 : the graph relaunched WITHOUT updating arguments (what it would cost if nothing changed between frames)
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/graph_vs_launch scripts/micro/graph_vs_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_rank(float *out, float camera) { out[threadIdx.x] = camera + (float)threadIdx.x; }
__global__ void k_assemble(const float *const *parts, int n, float *frame)
{
    float s = 0;
    for (int r = 0; r < n; r++) s += parts[r][threadIdx.x];
    frame[threadIdx.x] = s;
}

int main()
{
    const int FRAMES = 2000, WARM = 200;
    for (int N : { 1, 2, 4, 8 }) {
        std::vector<hipStream_t> st(N + 1);
        std::vector<hipEvent_t> ev(N);
        std::vector<float *> part(N);
        for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &p : part) CHECK(hipMalloc(&p, 64 * sizeof(float)));
        float **d_parts, *frame;
        CHECK(hipMalloc(&d_parts, N * sizeof(float *)));
        CHECK(hipMemcpy(d_parts, part.data(), N * sizeof(float *), hipMemcpyHostToDevice));
        CHECK(hipMalloc(&frame, 64 * sizeof(float)));
        hipStream_t s0 = st[N];

        auto direct = [&](float cam) -> int {
            for (int r = 0; r < N; r++) {
                hipLaunchKernelGGL(k_rank, dim3(1), dim3(64), 0, st[r], part[r], cam);
                CHECK(hipEventRecord(ev[r], st[r]));
                CHECK(hipStreamWaitEvent(s0, ev[r], 0));
            }
            hipLaunchKernelGGL(k_assemble, dim3(1), dim3(64), 0, s0, (const float *const *)d_parts, N, frame);
            CHECK(hipStreamSynchronize(s0));
            return 0;
        };
        double t_direct = 0;
        for (int f = 0; f < FRAMES + WARM; f++) {
            if (f == WARM) t_direct = -std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
            if (direct((float)f)) return 1;
        }
        t_direct += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();

        // the same frame as a graph: fork from s0 into the rank streams, join back
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipEvent_t fork;
        CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        CHECK(hipEventRecord(fork, s0));
        for (int r = 0; r < N; r++) {
            CHECK(hipStreamWaitEvent(st[r], fork, 0));
            hipLaunchKernelGGL(k_rank, dim3(1), dim3(64), 0, st[r], part[r], 0.0f);
            CHECK(hipEventRecord(ev[r], st[r]));
            CHECK(hipStreamWaitEvent(s0, ev[r], 0));
        }
        hipLaunchKernelGGL(k_assemble, dim3(1), dim3(64), 0, s0, (const float *const *)d_parts, N, frame);
        CHECK(hipStreamEndCapture(s0, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        size_t nn = 0;
        CHECK(hipGraphGetNodes(graph, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn), rank_nodes;
        CHECK(hipGraphGetNodes(graph, nodes.data(), &nn));
        for (auto nd : nodes) {
            hipGraphNodeType ty;
            CHECK(hipGraphNodeGetType(nd, &ty));
            if (ty != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams kp;
            CHECK(hipGraphKernelNodeGetParams(nd, &kp));
            if (kp.func == (void *)k_rank) rank_nodes.push_back(nd);
        }
        if ((int)rank_nodes.size() != N) { printf("found %zu rank nodes of %d\n", rank_nodes.size(), N); return 1; }
        double t_graph[2] = { 0, 0 };
        for (int update = 1; update >= 0; update--) {
            for (int f = 0; f < FRAMES + WARM; f++) {
                if (f == WARM) t_graph[update] = -std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
                if (update)
                    for (int r = 0; r < N; r++) {
                        hipKernelNodeParams kp;
                        CHECK(hipGraphKernelNodeGetParams(rank_nodes[r], &kp));
                        float cam = (float)f;
                        float *out = *(float **)kp.kernelParams[0];
                        void *args[2] = { &out, &cam };
                        kp.kernelParams = args;
                        CHECK(hipGraphExecKernelNodeSetParams(exec, rank_nodes[r], &kp));
                    }
                CHECK(hipGraphLaunch(exec, s0));
                CHECK(hipStreamSynchronize(s0));
            }
            t_graph[update] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
        }
        float h[64];
        CHECK(hipMemcpy(h, frame, sizeof h, hipMemcpyDeviceToHost));
        printf("%d rank stream(s): direct %.1f us per frame, graph with %d argument updates %.1f us, graph relaunched unchanged %.1f us   (check %.0f)\n",
               N, t_direct / FRAMES, N, t_graph[1] / FRAMES, t_graph[0] / FRAMES, h[1]);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
        for (auto &s : st) CHECK(hipStreamDestroy(s));
    }
    return 0;
}
