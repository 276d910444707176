// launch_chain.hip -- what does one dependent kernel launch cost on this box?  Chains of N trivial kernels (one wave / 256
// workgroups x 256 threads) in one stream, eagerly and replayed from a hipGraph; and the same chain split over 2 / 4 streams with
// fork / join events every 8 kernels (the HRNet module pattern).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_chain.hip -o scripts/micro/_bin/launch_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float* buf;
    hipMalloc(&buf, 4 << 20);
    hipMemset(buf, 0, 4 << 20);
    hipStream_t st, side[3];
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (auto& s : side) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t fork, join[3];
    hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    for (auto& e : join) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 320;
    for (int wgs : {1, 256, 1024}) {
        for (int branches : {1, 2, 4}) {
            auto enqueue = [&]() {
                // N kernels in total; with branches > 1, modules of 8 kernels per branch run on side streams between fork / join
                int left = N;
                while (left > 0) {
                    if (branches == 1) { hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, buf, wgs * 256); --left; continue; }
                    hipEventRecord(fork, st);
                    for (int b = 1; b < branches; ++b) hipStreamWaitEvent(side[b - 1], fork, 0);
                    for (int k = 0; k < 8; ++k)
                        for (int b = 0; b < branches; ++b) {
                            hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, b ? side[b - 1] : st, buf + b * (1 << 18), wgs * 256);
                            --left;
                        }
                    for (int b = 1; b < branches; ++b) { hipEventRecord(join[b - 1], side[b - 1]); hipStreamWaitEvent(st, join[b - 1], 0); }
                }
            };
            for (int graph = 0; graph < 2; ++graph) {
                hipGraphExec_t ge = nullptr;
                if (graph) {
                    hipGraph_t g;
                    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
                    enqueue();
                    hipStreamEndCapture(st, &g);
                    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                    hipGraphDestroy(g);
                }
                double best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    hipStreamSynchronize(st);
                    const double t0 = now();
                    if (graph) hipGraphLaunch(ge, st); else enqueue();
                    hipStreamSynchronize(st);
                    const double dt = now() - t0;
                    if (rep && dt < best) best = dt;
                }
                printf("wgs %4d branches %d %s: %7.1f us total, %5.2f us per kernel (%d kernels)\n", wgs, branches, graph ? "graph" : "eager", best * 1e6,
                       best * 1e6 / N, N);
                if (ge) hipGraphExecDestroy(ge);
            }
        }
    }
    return 0;
}
