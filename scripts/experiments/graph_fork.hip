// Experiment (round 3): does a replayed hipGraph run two branches of a fork concurrently on this runtime, i.e. can a tiny
// "decide" kernel hide next to the backbone launch instead of sitting in the dependent chain?
//   serial     : R x [A, C]                 (the plain loop: step, backbone)
//   serial+B   : R x [A, B, C]              (a decision kernel in the chain)
//   fork/join  : R x [A, {B || C}, join]    (B on a second capture stream)
// build: hipcc --offload-arch=gfx950 -O2 scripts/experiments/graph_fork.hip -o build/graph_fork
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

__global__ void big(float* p, int n) {                       // 65536-element elementwise kernel (step / backbone stand-in)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 0.999f + 1.0f;
}
__global__ void tiny(float* q) {                              // one wave (decision kernel stand-in)
    if (threadIdx.x == 0) q[0] += 1.0f;
}
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main() {
    const int n = 65536, R = 10;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *a, *c, *q;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&q, 256));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(c, 0, n * 4)); CK(hipMemset(q, 0, 256));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    auto build = [&](int mode, hipGraphExec_t* ex) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < R; ++r) {
            hipLaunchKernelGGL(big, dim3(n / 256), dim3(256), 0, s1, a, n);                 // A: step
            if (mode == 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, q);           // B in the chain
            if (mode == 2) {                                                               // B on a forked branch
                CK(hipEventRecord(ef, s1));
                CK(hipStreamWaitEvent(s2, ef, 0));
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, q);
                CK(hipEventRecord(ej, s2));
            }
            hipLaunchKernelGGL(big, dim3(n / 256), dim3(256), 0, s1, c, n);                 // C: backbone
            if (mode == 2) CK(hipStreamWaitEvent(s1, ej, 0));                               // join before the next step
        }
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(ex, g, nullptr, nullptr, 0));
    };
    const char* names[3] = {"serial     R x [A, C]", "serial + B R x [A, B, C]", "fork/join  R x [A, {B || C}]"};
    for (int mode = 0; mode < 3; ++mode) {
        hipGraphExec_t ex;
        build(mode, &ex);
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ex, s1));
        CK(hipStreamSynchronize(s1));
        const int reps = 1000;
        const double t0 = now_us();
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ex, s1));
        CK(hipStreamSynchronize(s1));
        const double per = (now_us() - t0) / reps;
        std::printf("%-32s %.2f us per graph launch, %.2f us per iteration\n", names[mode], per, per / R);
    }
    return 0;
}
