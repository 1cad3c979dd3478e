// valu_mix_calib.hip — cycles one gfx950 SIMD spends per wave64 instruction, by KIND of instruction, at the occupancy the FIM
// worker runs at (4 waves per SIMD) and at 8.  tools/valu_calib.hip settled the figure for v_fma_f32; the FIM worker's
// scoring call is mostly integer / select / compare work, so its roof depends on what those kinds cost.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_mix_calib tools/valu_mix_calib.hip && ./tools/valu_mix_calib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 16;
constexpr int kIters = 2048;

enum Kind { FMA, PK_FMA, XOR, CNDMASK, MIN_U32, CMP, MAD_U24, LSHL_ADD, ADD3, MUL_F64, RCP, MUL_LO, SALU, FMA_SALU, XOR_SALU,
            FMA_S, MUL_S, XOR_S, CND_E64, CMP_E64, MIN_E32, MIN3, AND, ADD_U32, LSHLREV, CVT, RNDNE, MAX3, SUB_F32, MUL_F32, NOP, WAITCNT, S_MOV, S_ADD, BFE, PAIR_VCC, PAIR_SGPR, PAIR_VCC_NOP, CND_VCC_AFTER_VALU, READLANE, KINDS };
static const char *kNames[KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_xor_b32", "v_cndmask_b32", "v_min_u32", "v_cmp_gt_u32 (vcc)", "v_mad_u32_u24",
                                    "v_lshl_add_u32", "v_add3_u32", "v_mul_f64", "v_rcp_f32", "v_mul_lo_u32", "s_and_b64", "v_fma_f32 + s_and_b64 alternating (per pair)",
                                    "v_xor_b32 + s_and_b64 alternating (per pair)",
                                    "v_fma_f32 v, s, v, v", "v_mul_f32_e32 v, s, v", "v_xor_b32_e32 v, s, v", "v_cndmask_b32_e64 (sgpr pair)", "v_cmp_gt_u32_e64 (sgpr pair)",
                                    "v_min_u32_e32", "v_min3_u32", "v_and_b32_e32", "v_add_u32_e32", "v_lshlrev_b32_e32", "v_cvt_i32_f32_e32", "v_rndne_f32_e32",
                                    "v_max3_f32", "v_sub_f32_e32", "v_mul_f32_e32", "s_nop 0", "s_waitcnt lgkmcnt(0) (nothing pending)", "s_mov_b32", "s_add_i32", "v_bfe_u32",
                                    "v_cmp_e32 vcc + v_cndmask_e32 vcc (per pair)", "v_cmp_e64 sgpr + v_cndmask_e64 sgpr (per pair)", "v_cmp_e32 vcc + s_nop 1 + v_cndmask_e32 vcc (per triple)",
                                    "v_cndmask_b32_e32 vcc, vcc last written by a v_cmp before the loop", "v_readlane_b32"};

template <int K>
__device__ __forceinline__ void op(float &x, float a, float b, double &d, unsigned long long &sm)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    if (K == FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if (K == PK_FMA) { f2 &p = reinterpret_cast<f2 &>(d); asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p) : "v"(p)); }
    if (K == XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
    if (K == MIN_U32) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == CMP) asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(x), "v"(a) : "vcc");
    if (K == MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if (K == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(x) : "v"(a));
    if (K == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if (K == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(d));
    if (K == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (K == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == SALU) asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm) :: "scc");
    if (K == FMA_SALU) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm) :: "scc"); }
    if (K == FMA_S) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "s"(a), "v"(b));
    if (K == MUL_S) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "s"(a));
    if (K == XOR_S) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x) : "s"(a));
    if (K == CND_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(sm));
    if (K == CMP_E64) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(sm) : "v"(x), "v"(a));
    if (K == MIN_E32) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == MIN3) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if (K == AND) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == ADD_U32) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == LSHLREV) asm volatile("v_lshlrev_b32_e32 %0, 2, %0" : "+v"(x));
    if (K == CVT) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(x));
    if (K == RNDNE) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(x));
    if (K == MAX3) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if (K == SUB_F32) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == MUL_F32) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
    if (K == NOP) asm volatile("s_nop 0");
    if (K == WAITCNT) asm volatile("s_waitcnt lgkmcnt(0)");
    if (K == S_MOV) { unsigned lo; asm volatile("s_mov_b32 %0, 7" : "=s"(lo)); }
    if (K == S_ADD) { unsigned &lo = reinterpret_cast<unsigned &>(sm); asm volatile("s_add_i32 %0, %0, 3" : "+s"(lo) :: "scc"); }
    if (K == BFE) asm volatile("v_bfe_u32 %0, %0, %1, 2" : "+v"(x) : "v"(a));
    if (K == PAIR_VCC) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
    if (K == PAIR_SGPR) asm volatile("v_cmp_gt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(sm));
    if (K == PAIR_VCC_NOP) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
    if (K == CND_VCC_AFTER_VALU) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a));
    if (K == READLANE) { unsigned lo; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(lo) : "v"(x)); }
    if (K == XOR_SALU) { asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(a)); asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm) :: "scc"); }
}

template <int W, int K>
__global__ __launch_bounds__(256) void stream(float *out, unsigned long long *cycles, float a, float b)
{
    extern __shared__ float pin[];
    if (a == 12345.f) pin[threadIdx.x] = b;
    float acc[kUnroll];
    double dd[kUnroll];
    unsigned long long sm[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) { acc[u] = (float)(threadIdx.x + u); dd[u] = 1.0 + 1e-9 * u; sm[u] = ~0ull; }
    unsigned long long t0, t1;
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    if (K == CND_VCC_AFTER_VALU) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" :: "v"(acc[0]), "v"(acc[1]) : "vcc");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) op<K>(acc[u], a, b, dd[u], sm[u]);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) s += acc[u] + (float)dd[u] + (float)(sm[u] & 1ull);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int W, int K>
int run(int cus, float *d_out, unsigned long long *d_cyc)
{
    const int blocks = cus * W;
    const size_t lds = (size_t)(160 * 1024) / W - 4096;
    auto kern = stream<W, K>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, d_cyc, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, d_cyc, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[cyc.size() / 2], insts = (double)kUnroll * kIters;
    std::printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.3f, \"simd_cycles_per_wave_inst\": %.3f}\n",
                kNames[K], W, med / insts, med / insts / W);
    std::fflush(stdout);
    return 0;
}

template <int K>
int both(int cus, float *d_out, unsigned long long *d_cyc)
{
    if (run<1, K>(cus, d_out, d_cyc)) return 1;
    if (run<4, K>(cus, d_out, d_cyc)) return 1;
    return run<8, K>(cus, d_out, d_cyc);
}

int main()
{
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *d_out; unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
    if (both<FMA>(cus, d_out, d_cyc) || both<PK_FMA>(cus, d_out, d_cyc) || both<XOR>(cus, d_out, d_cyc) || both<CNDMASK>(cus, d_out, d_cyc) ||
        both<MIN_U32>(cus, d_out, d_cyc) || both<CMP>(cus, d_out, d_cyc) || both<MAD_U24>(cus, d_out, d_cyc) || both<LSHL_ADD>(cus, d_out, d_cyc) ||
        both<ADD3>(cus, d_out, d_cyc) || both<MUL_F64>(cus, d_out, d_cyc) || both<RCP>(cus, d_out, d_cyc) || both<MUL_LO>(cus, d_out, d_cyc) ||
        both<SALU>(cus, d_out, d_cyc) || both<FMA_SALU>(cus, d_out, d_cyc) || both<XOR_SALU>(cus, d_out, d_cyc)) return 1;
    if (both<FMA_S>(cus, d_out, d_cyc) || both<MUL_S>(cus, d_out, d_cyc) || both<XOR_S>(cus, d_out, d_cyc) || both<CND_E64>(cus, d_out, d_cyc) ||
        both<CMP_E64>(cus, d_out, d_cyc) || both<MIN_E32>(cus, d_out, d_cyc) || both<MIN3>(cus, d_out, d_cyc) || both<AND>(cus, d_out, d_cyc) ||
        both<ADD_U32>(cus, d_out, d_cyc) || both<LSHLREV>(cus, d_out, d_cyc) || both<CVT>(cus, d_out, d_cyc) || both<RNDNE>(cus, d_out, d_cyc) ||
        both<MAX3>(cus, d_out, d_cyc) || both<SUB_F32>(cus, d_out, d_cyc) || both<MUL_F32>(cus, d_out, d_cyc) || both<NOP>(cus, d_out, d_cyc) ||
        both<WAITCNT>(cus, d_out, d_cyc) || both<S_MOV>(cus, d_out, d_cyc) || both<S_ADD>(cus, d_out, d_cyc) || both<BFE>(cus, d_out, d_cyc)) return 1;
    if (both<PAIR_VCC>(cus, d_out, d_cyc) || both<PAIR_SGPR>(cus, d_out, d_cyc) || both<PAIR_VCC_NOP>(cus, d_out, d_cyc) || both<CND_VCC_AFTER_VALU>(cus, d_out, d_cyc) ||
        both<READLANE>(cus, d_out, d_cyc)) return 1;
    return 0;
}
