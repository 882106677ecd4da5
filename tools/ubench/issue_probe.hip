// issue_probe.hip -- how fast does ONE wave (and 2, 3 waves per SIMD) issue fp64 VALU work on gfx950?
// Measures cycles per instruction (s_memtime) of hand-written instruction streams:
//   dep      one dependent chain of v_fma_f64
//   ind2/4   2 / 4 independent chains, interleaved
//   lit      dependent chain where every FMA takes its addend from an SGPR pair filled by two s_mov_b32 just before
//   litind2  the same with two interleaved chains
//   salu     dependent chain with 1 unrelated s_mov_b32 per FMA
// build: hipcc -O3 --offload-arch=gfx950 issue_probe.hip -o issue_probe ; run: ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters, double x0)
{
    double a = x0 + threadIdx.x * 1e-9, b = a + 1., c = a + 2., d = a + 3.;
    const double m = 0.999999;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(m));)
        } else if (MODE == 1) {
            REP64(asm volatile("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2" : "+v"(a), "+v"(b) : "v"(m));)
        } else if (MODE == 2) {
            REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
        } else if (MODE == 3) {
            REP64(asm volatile("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3fe12345\n v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(m) : "s20", "s21");)
        } else if (MODE == 4) {
            REP64(asm volatile("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3fe12345\n v_fma_f64 %0, %0, %2, s[20:21]\n"
                               "s_mov_b32 s22, 0x12345679\n s_mov_b32 s23, 0x3fe12346\n v_fma_f64 %1, %1, %2, s[22:23]" : "+v"(a), "+v"(b) : "v"(m) : "s20", "s21", "s22", "s23");)
        } else if (MODE == 5) {
            REP64(asm volatile("s_mov_b32 s20, 0x12345678\n v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(m) : "s20");)
        } else if (MODE == 6) {   // fp64 mul/add mix, independent pairs
            REP64(asm volatile("v_mul_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));)
        } else if (MODE == 7) {   // v_rcp_f64 chain
            REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
        } else if (MODE == 9) {   // dependent chain of DPP-broadcast fmacs (constant from lane k of each 16-lane row)
            REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(m));)
        } else if (MODE == 10) {  // two independent accumulators
            REP64(asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c) : "v"(b), "v"(m));)
        } else if (MODE == 11) {  // v_mov_b64_dpp + dependent fma
            REP64(asm volatile("v_mov_b64_dpp %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %0, %3, %1" : "+v"(a), "=&v"(c) : "v"(b), "v"(m));)
        } else if (MODE == 12) {  // plain v_fmac_f64 (VOP2) chain
            REP64(asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a) : "v"(b), "v"(m));)
        } else if (MODE == 8) {   // dependent fma + v_cndmask pair (32-bit ops)
            REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_mov_b32 %2, %2" : "+v"(a) : "v"(m), "v"(threadIdx.x));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_rep, int waves_per_simd)
{
    const int iters = 200;
    // one workgroup of 256 threads = 4 waves = one wave per SIMD of a CU; waves_per_simd workgroups per CU
    const int blocks = 256 * waves_per_simd;
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += double(v);
    s /= h.size();
    const double n = double(iters) * 64 * instr_per_rep;
    printf("%-8s waves/SIMD %d: %.2f counter ticks per instruction per wave (%d instr/rep); kernel %.3f ms -> %.2f ns per instr per wave\n",
           name, waves_per_simd, s / n, instr_per_rep, ms, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

__global__ void k_check(double *o, const double *c)
{
    const double tab = c[threadIdx.x & 15];
    double acc;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(acc) : "v"(tab));
    const double x = double(threadIdx.x);
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x));
    o[threadIdx.x] = acc;   // expect c[3] + c[5]*lane
}

// does a DPP broadcast deliver data from a lane that EXEC has switched off?
__global__ void k_check_exec(double *o, const double *c)
{
    const double tab = c[threadIdx.x & 15];
    double acc = -1.;
    if (threadIdx.x & 1) {       // odd lanes only; lane 4 of every row is inactive
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab));
    }
    o[threadIdx.x] = acc;
}
__global__ void k_check_exec_bc(double *o, const double *c)
{
    const double tab = c[threadIdx.x & 15];
    double acc = -1.;
    if (threadIdx.x & 1) {
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(tab));
    }
    o[threadIdx.x] = acc;
}

int main()
{
    {
        double hc[16], ho[64], *dc, *dout;
        for (int i = 0; i < 16; ++i) hc[i] = 100. + i;
        hipMalloc(&dc, sizeof hc); hipMalloc(&dout, sizeof ho);
        hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check_exec, dim3(1), dim3(64), 0, 0, dout, dc);
        hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        printf("DPP source lane disabled by EXEC, bound_ctrl:0 -> odd lane gets %g (104 = data delivered, -1 = write suppressed, 0 = zero)\n", ho[17]);
        hipLaunchKernelGGL(k_check_exec_bc, dim3(1), dim3(64), 0, 0, dout, dc);
        hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        printf("DPP source lane disabled by EXEC, bound_ctrl:1 -> odd lane gets %g\n", ho[17]);
    }
    {
        double hc[16], ho[64], *dc, *dout;
        for (int i = 0; i < 16; ++i) hc[i] = 100. + i;
        hipMalloc(&dc, sizeof hc); hipMalloc(&dout, sizeof ho);
        hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dout, dc);
        hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; ++i) if (ho[i] != 103. + 105. * i) ++bad;
        printf("row_newbcast check: %d of 64 lanes wrong (lane 17 got %g, want %g)\n", bad, ho[17], 103. + 105. * 17);
    }
    for (int w = 1; w <= 3; ++w) {
        run<0>("dep", 1, w);
        run<1>("ind2", 2, w);
        run<2>("ind4", 4, w);
        run<3>("lit", 3, w);
        run<4>("litind2", 6, w);
        run<5>("salu", 2, w);
        run<6>("muladd", 2, w);
        run<7>("rcp", 1, w);
        run<8>("fma+mov", 2, w);
        run<9>("fmacdpp", 1, w);
        run<10>("fmacdpp2", 2, w);
        run<11>("movdpp+fma", 2, w);
        run<12>("fmac", 1, w);
    }
    return 0;
}
