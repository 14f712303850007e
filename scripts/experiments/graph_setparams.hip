// Experiment (round 3): can the per-call pointers of the FIRST node of a replayed hipGraph be refreshed with
// hipGraphExecKernelNodeSetParams instead of an eager launch in front of hipGraphLaunch?
//   A. semantics: with the GPU held busy, enqueue N x (SetParams -> hipGraphLaunch) back to back; every launch must
//      see ITS params (CUDA's contract: updates affect future launches only).  An in-place rewrite of a shared
//      kernarg buffer would show up as launches seeing a later launch's params.
//   B. cost: host us per (SetParams + launch) and GPU us per call for
//        graph only | eager kernel + graph | SetParams + graph (kernel inside)
// build: hipcc --offload-arch=gfx950 -O2 scripts/experiments/graph_setparams.hip -o build/graph_setparams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

__global__ void spin(unsigned long long ticks) {            // wall_clock64: 100 MHz
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
struct Desc { int* dst; int val; int pad[30]; };              // a descriptor-sized by-value argument like lp_step_desc
__global__ void writer(int* dst, int val, const Desc d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { dst[0] = val; d.dst[64] = d.val; }
}
__global__ void chain(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 0.999f + 1.0f;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int n_chain = argc > 1 ? std::atoi(argv[1]) : 12;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* buf; float* data;
    const int N = 64, n_el = 65536;
    CK(hipMalloc(&buf, 2 * N * sizeof(int)));
    CK(hipMalloc(&data, n_el * sizeof(float)));
    CK(hipMemset(buf, 0xff, 2 * N * sizeof(int)));
    CK(hipMemset(data, 0, n_el * sizeof(float)));

    auto capture = [&](bool with_writer, hipGraph_t* g, hipGraphExec_t* ex) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        if (with_writer) { Desc d{buf, -7, {}}; hipLaunchKernelGGL(writer, dim3(1), dim3(64), 0, s, buf, -7, d); }
        for (int k = 0; k < n_chain; ++k) hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
        CK(hipStreamEndCapture(s, g));
        CK(hipGraphInstantiate(ex, *g, nullptr, nullptr, 0));
    };
    hipGraph_t g_in, g_out; hipGraphExec_t e_in, e_out;
    capture(true, &g_in, &e_in);
    capture(false, &g_out, &e_out);

    // the root node of g_in is the writer launch
    size_t n_root = 0;
    CK(hipGraphGetRootNodes(g_in, nullptr, &n_root));
    std::vector<hipGraphNode_t> roots(n_root);
    CK(hipGraphGetRootNodes(g_in, roots.data(), &n_root));
    hipGraphNodeType ty;
    CK(hipGraphNodeGetType(roots[0], &ty));
    hipKernelNodeParams base{};
    CK(hipGraphKernelNodeGetParams(roots[0], &base));
    std::printf("roots %zu type %d (kernel = %d) func match %d grid %u block %u\n", n_root, (int)ty, (int)hipGraphNodeTypeKernel,
                (int)(base.func == (void*)writer), base.gridDim.x, base.blockDim.x);

    auto set = [&](int i) {
        int* dst = buf + i; int val = i; Desc d{buf + i, 1000 + i, {}};
        void* args[3] = {&dst, &val, &d};
        hipKernelNodeParams p = base;
        p.kernelParams = args; p.extra = nullptr;
        CK(hipGraphExecKernelNodeSetParams(e_in, roots[0], &p));
    };

    // ---- A: semantics under a busy GPU ------------------------------------------------------------------
    for (int round = 0; round < 3; ++round) {
        CK(hipMemset(buf, 0xff, 2 * N * sizeof(int)));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 3000000ull);     // 30 ms
        const double t0 = now_us();
        for (int i = 0; i < N; ++i) { set(i); CK(hipGraphLaunch(e_in, s)); }
        const double t_enq = now_us() - t0;
        CK(hipStreamSynchronize(s));
        std::vector<int> h(2 * N);
        CK(hipMemcpy(h.data(), buf, 2 * N * sizeof(int), hipMemcpyDeviceToHost));
        int bad = 0, bad_d = 0;
        for (int i = 0; i < N; ++i) { bad += h[i] != i; bad_d += h[64 + i] != 1000 + i; }
        std::printf("A round %d: %d launches enqueued in %.0f us behind a 30 ms kernel; scalar-arg mismatches %d, by-value-desc mismatches %d  (first: %d %d %d ... last: %d)\n",
                    round, N, t_enq, bad, bad_d, h[0], h[1], h[2], h[N - 1]);
    }

    // ---- B: cost -----------------------------------------------------------------------------------------
    const int reps = 2000;
    auto timeit = [&](const char* name, auto&& body) {
        for (int i = 0; i < 50; ++i) body(i);
        CK(hipStreamSynchronize(s));
        const double t0 = now_us();
        for (int i = 0; i < reps; ++i) body(i);
        const double t_host = now_us() - t0;
        CK(hipStreamSynchronize(s));
        const double t_all = now_us() - t0;
        std::printf("B %-44s host %.2f us/call, wall %.2f us/call\n", name, t_host / reps, t_all / reps);
    };
    timeit("graph only (no first kernel)", [&](int) { CK(hipGraphLaunch(e_out, s)); });
    timeit("eager kernel + graph", [&](int i) {
        Desc d{buf + (i & 63), i, {}};
        hipLaunchKernelGGL(writer, dim3(1), dim3(64), 0, s, buf + (i & 63), i, d);
        CK(hipGraphLaunch(e_out, s));
    });
    timeit("SetParams + graph (kernel is node 0)", [&](int i) { set(i & 63); CK(hipGraphLaunch(e_in, s)); });
    timeit("graph with node 0, params untouched", [&](int) { CK(hipGraphLaunch(e_in, s)); });
    // an eager elementwise kernel between graph launches, as the sampler's Euler update is
    timeit("eager chain + SetParams + graph", [&](int i) {
        hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
        set(i & 63); CK(hipGraphLaunch(e_in, s));
    });
    timeit("eager chain + eager kernel + graph", [&](int i) {
        hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
        Desc d{buf + (i & 63), i, {}};
        hipLaunchKernelGGL(writer, dim3(1), dim3(64), 0, s, buf + (i & 63), i, d);
        CK(hipGraphLaunch(e_out, s));
    });
    // ---- C: is the fixed cost of a graph launch tied to re-launching the SAME exec?  Alternate two / four instances of the
    // same graph; and a graph launched every time on a freshly alternating pair of streams is NOT what a sigma call can do
    // (stream order is needed), so only the exec is varied.
    {
        hipGraphExec_t ex[4];
        for (int k = 0; k < 4; ++k) CK(hipGraphInstantiate(&ex[k], g_out, nullptr, nullptr, 0));
        timeit("graph only, 2 exec instances alternating", [&](int i) { CK(hipGraphLaunch(ex[i & 1], s)); });
        timeit("graph only, 4 exec instances rotating", [&](int i) { CK(hipGraphLaunch(ex[i & 3], s)); });
        timeit("eager chain + graph, 2 execs alternating", [&](int i) {
            hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
            CK(hipGraphLaunch(ex[i & 1], s));
        });
        timeit("eager chain + graph, same exec", [&](int i) {
            hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
            CK(hipGraphLaunch(e_out, s));
        });
    }
    // ---- D: the same 12 kernels launched eagerly from this C loop (no graph at all)
    timeit("12 eager chain launches (no graph)", [&](int) {
        for (int k = 0; k < n_chain; ++k) hipLaunchKernelGGL(chain, dim3(n_el / 256), dim3(256), 0, s, data, n_el);
    });
    std::printf("done\n");
    return 0;
}
