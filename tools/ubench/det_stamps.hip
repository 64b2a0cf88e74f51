// Development aid: the long-frame detector (long_phases.h) with a clock stamp after every phase of one workgroup,
// on random data with a zeroed blind zone, all compute units busy.  Prints cycles (100 MHz wall clock) per phase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../dsp_amd/csrc/long_phases.h"
using namespace echo;

template <int NT>
__global__ __launch_bounds__(NT) void det_stamped(LongParams p, long long* stamps, int* nstamps) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    int ns = 0;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads(); if (tid == 0 && blk == 700 && ns < 200) stamps[ns++] = wall_clock64();
    if (tid == 0 && blk == 700) stamps[ns++] = wall_clock64();
    ECHO_LDET_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, NT);
    if (tid == 0 && blk == 700) { stamps[ns++] = wall_clock64(); *nstamps = ns; }
}

template <int NT> void run(LongParams p, int nblk, long long* st, int* ns) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(det_stamped<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, p.det_smem);
    for (int it = 0; it < 2; ++it) det_stamped<NT><<<nblk, NT, p.det_smem>>>(p, st, ns);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); det_stamped<NT><<<nblk, NT, p.det_smem>>>(p, st, ns); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int n; long long s[200];
    hipMemcpy(&n, ns, 4, hipMemcpyDeviceToHost); hipMemcpy(s, st, sizeof(s), hipMemcpyDeviceToHost);
    printf("NT %d: kernel %.3f ms for %d blocks; %d stamps; block total %lld ticks (10 ns each)\n", NT, ms, nblk, n, s[n - 1] - s[0]);
    for (int i = 1; i < n; ++i) printf("%lld ", s[i] - s[i - 1]);
    printf("\n");
}

int main() {
    LongParams p; std::memset(&p, 0, sizeof(p));
    if (!plan_long(p, 8820, 4410, 101, 3, 500, 1)) return 1;
    const int nblk = 6144, npos = p.L - p.half;
    std::vector<double> h((size_t)nblk * npos);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(g);
    double* d; BandRes* r; long long* st; int* ns;
    hipMalloc(&d, h.size() * 8); hipMalloc(&r, nblk * sizeof(BandRes)); hipMalloc(&st, 200 * 8); hipMalloc(&ns, 4);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    p.det_io = d; p.det_stride = npos; p.det_off = 0; p.write_mutated = 0; p.res_out = r;
    run<1024>(p, nblk, st, ns);
    run<512>(p, nblk, st, ns);
    p.det_smem += 30000;                      // one workgroup per compute unit, for comparison
    run<1024>(p, nblk, st, ns);
    return 0;
}
