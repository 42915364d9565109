// tools/ubench_valu.hip - issue-rate calibration for the instructions the scan kernel is built from.
// Not part of the product: a measuring aid (DESIGN.md quotes its output).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
// Each kernel runs ITER iterations of 8 independent chains of one instruction, on `waves` wavefronts per
// SIMD of every CU; prints SIMD-cycles per wave-instruction assuming the clock given by wall time of a
// reference v_add_u32 kernel is what it is (so ratios are exact, absolute cycles use 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define BODY8(INS)                                                                                      \
    INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)

#define KERNEL(NAME, ASM_TEMPLATE)                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                         \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,  \
                 a6 = a0 * 17, a7 = a0 * 19;                                                            \
        uint32_t b = seed * 77 + threadIdx.x, c = seed ^ 0x12345;                                       \
        for (int i = 0; i < ITER / 4; i++) {                                                            \
            ASM_TEMPLATE(a0) ASM_TEMPLATE(a1) ASM_TEMPLATE(a2) ASM_TEMPLATE(a3)                          \
            ASM_TEMPLATE(a4) ASM_TEMPLATE(a5) ASM_TEMPLATE(a6) ASM_TEMPLATE(a7)                          \
            ASM_TEMPLATE(a0) ASM_TEMPLATE(a1) ASM_TEMPLATE(a2) ASM_TEMPLATE(a3)                          \
            ASM_TEMPLATE(a4) ASM_TEMPLATE(a5) ASM_TEMPLATE(a6) ASM_TEMPLATE(a7)                          \
            ASM_TEMPLATE(a0) ASM_TEMPLATE(a1) ASM_TEMPLATE(a2) ASM_TEMPLATE(a3)                          \
            ASM_TEMPLATE(a4) ASM_TEMPLATE(a5) ASM_TEMPLATE(a6) ASM_TEMPLATE(a7)                          \
            ASM_TEMPLATE(a0) ASM_TEMPLATE(a1) ASM_TEMPLATE(a2) ASM_TEMPLATE(a3)                          \
            ASM_TEMPLATE(a4) ASM_TEMPLATE(a5) ASM_TEMPLATE(a6) ASM_TEMPLATE(a7)                          \
        }                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
    }

#define I_ADD(x)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_AND(x)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_PKMAX(x)    asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_PKMIN(x)    asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_PKSUBS(x)   asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
#define I_PKADD(x)    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_PKMUL(x)    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_PKMAD(x)    asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_PERM(x)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_ALIGN(x)    asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x) : "v"(b));
#define I_MAX3(x)     asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MAXU16(x)   asm volatile("v_max_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MAX3U16(x)  asm volatile("v_max3_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_FMA(x)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_PKFMA(x)    asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MADU24(x)   asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_DOT4(x)     asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_LSHLOR(x)   asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define I_SADU8(x)    asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_CMP(x)      asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define I_MOVDPP(x)   asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define I_SUBSDWA(x)  asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(x) : "v"(b));

#define I_CMPVCC(x)   asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define I_CMPSG(x)    asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21");
#define I_CMPU16(x)   asm volatile("v_cmp_gt_u16 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define I_ADDC(x)     asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : : "vcc");
#define I_ADDCSG(x)   asm volatile("v_addc_co_u32 %0, s[22:23], %0, %0, s[20:21]" : "+v"(x) : : "s22", "s23");
#define I_MAXU32(x)   asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MINU32(x)   asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_SUBU32(x)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_LSHR(x)     asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(x));
#define I_OR(x)       asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_CNDMASK(x)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define I_MAXF32(x)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_SUBF32C(x)  asm volatile("v_sub_f32 %0, %0, %1 clamp" : "+v"(x) : "v"(b));
#define I_MULF32(x)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MACF32(x)   asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_FMAC(x)     asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x) : "v"(b), "v"(c));
#define I_CVTUB(x)    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x));
#define I_CVTU32(x)   asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define I_MULU24(x)   asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_BFI(x)      asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_BFE(x)      asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x));
#define I_ADD3(x)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_LSHLADD(x)  asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
#define I_ANDOR(x)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_SUBU16(x)   asm volatile("v_sub_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MULLOU16(x) asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MADU16(x)   asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_XOR(x)      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_MAXI32(x)   asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_DOT4I(x)    asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_DOT2(x)     asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_V_S1(x)     asm volatile("v_max_u32 %0, %0, %1\n s_and_b64 s[20:21], s[22:23], s[24:25]" : "+v"(x) : "v"(b) : "s20", "s21", "scc");
#define I_V_S2(x)     asm volatile("v_max_u32 %0, %0, %1\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_andn2_b64 s[26:27], s[22:23], s[24:25]" : "+v"(x) : "v"(b) : "s20", "s21", "s26", "s27", "scc");
#define I_CMPSG_S1(x) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1\n s_and_b64 s[26:27], s[22:23], s[24:25]" : : "v"(x), "v"(b) : "s20", "s21", "s26", "s27", "scc");
#define I_CMP_ADDC2(x) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_max_u32 %0, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");

#define I_CND64(x)    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b));
#define I_CNDNEW(x)   asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(b), "v"(c) : "vcc");
#define I_BPERM(x)    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(b));
#define I_SWIZ(x)     asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n s_waitcnt lgkmcnt(0)" : "+v"(x));
#define I_DPPXOR(x)   asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define I_DPPROR(x)   asm volatile("v_add_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define I_READLANE(x) asm volatile("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0" : "+v"(x) : : "s20");
KERNEL(k_cnd64, I_CND64)
KERNEL(k_cndnew, I_CNDNEW)
KERNEL(k_bperm, I_BPERM)
KERNEL(k_swiz, I_SWIZ)
KERNEL(k_dppxor, I_DPPXOR)
KERNEL(k_dppror, I_DPPROR)
KERNEL(k_readlane, I_READLANE)
KERNEL(k_cmpvcc, I_CMPVCC)
KERNEL(k_cmpsg, I_CMPSG)
KERNEL(k_cmpu16, I_CMPU16)
KERNEL(k_addc, I_ADDC)
KERNEL(k_addcsg, I_ADDCSG)
KERNEL(k_maxu32, I_MAXU32)
KERNEL(k_minu32, I_MINU32)
KERNEL(k_subu32, I_SUBU32)
KERNEL(k_lshr, I_LSHR)
KERNEL(k_or, I_OR)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_maxf32, I_MAXF32)
KERNEL(k_subf32c, I_SUBF32C)
KERNEL(k_mulf32, I_MULF32)
KERNEL(k_macf32, I_MACF32)
KERNEL(k_fmac, I_FMAC)
KERNEL(k_cvtub, I_CVTUB)
KERNEL(k_cvtu32, I_CVTU32)
KERNEL(k_mulu24, I_MULU24)
KERNEL(k_bfi, I_BFI)
KERNEL(k_bfe, I_BFE)
KERNEL(k_add3, I_ADD3)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_subu16, I_SUBU16)
KERNEL(k_mullou16, I_MULLOU16)
KERNEL(k_madu16, I_MADU16)
KERNEL(k_xor, I_XOR)
KERNEL(k_maxi32, I_MAXI32)
KERNEL(k_dot4i, I_DOT4I)
KERNEL(k_dot2, I_DOT2)
KERNEL(k_v_s1, I_V_S1)
KERNEL(k_v_s2, I_V_S2)
KERNEL(k_cmpsg_s1, I_CMPSG_S1)
KERNEL(k_cmp_addc2, I_CMP_ADDC2)

KERNEL(k_add, I_ADD)
KERNEL(k_and, I_AND)
KERNEL(k_pkmax, I_PKMAX)
KERNEL(k_pkmin, I_PKMIN)
KERNEL(k_pksubs, I_PKSUBS)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_pkmul, I_PKMUL)
KERNEL(k_pkmad, I_PKMAD)
KERNEL(k_perm, I_PERM)
KERNEL(k_align, I_ALIGN)
KERNEL(k_max3, I_MAX3)
KERNEL(k_maxu16, I_MAXU16)
KERNEL(k_max3u16, I_MAX3U16)
KERNEL(k_fma, I_FMA)
KERNEL(k_pkfma16, I_PKFMA)
KERNEL(k_madu24, I_MADU24)
KERNEL(k_dot4, I_DOT4)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_sadu8, I_SADU8)
KERNEL(k_cmp_addc, I_CMP)
KERNEL(k_movdpp, I_MOVDPP)
KERNEL(k_subsdwa, I_SUBSDWA)

// LDS: b128 write + 3 b128 reads per iteration, like the scan ring
__global__ __launch_bounds__(256) void k_lds_ring(uint32_t *out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 v = make_uint4(seed, seed * 3, seed * 5, lane);
    uint32_t acc = 0;
    for (int i = 0; i < ITER; i++) {
        *reinterpret_cast<uint4 *>(&ring[wave][((i & 1) * 256) + lane * 4]) = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t d0 = (uint32_t)(((i & 1) * 256) + 512 - 8 + lane * 4) & 511u;
        const uint4 a = *reinterpret_cast<const uint4 *>(&ring[wave][d0]);
        const uint4 b = *reinterpret_cast<const uint4 *>(&ring[wave][(d0 + 4) & 511u]);
        const uint4 d = *reinterpret_cast<const uint4 *>(&ring[wave][(d0 + 8) & 511u]);
        acc += a.x ^ b.y ^ d.z;
        v.x += acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// neighbour exchange without LDS: 3 DPP/permlane-style moves per dword
__global__ __launch_bounds__(256) void k_dpp_shift(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    for (int i = 0; i < ITER; i++) {
        a0 += __builtin_amdgcn_update_dpp(0u, a1, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
        a1 += __builtin_amdgcn_update_dpp(0u, a2, 0x130, 0xf, 0xf, false);
        a2 += __builtin_amdgcn_update_dpp(0u, a3, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        a3 += __builtin_amdgcn_update_dpp(0u, a0, 0x138, 0xf, 0xf, false);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

// streaming read ceiling: 16 B per lane, grid-stride, nothing else
__global__ __launch_bounds__(256) void k_stream(const uint4 *in, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w) + (c.x ^ c.y ^ c.z ^ c.w) + (d.x ^ d.y ^ d.z ^ d.w);
    }
    for (; i < n16; i += stride) { const uint4 a = in[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// same, contiguous run per wavefront (the scan kernel's access pattern): wave w reads chunks [w*R, (w+1)*R)
__global__ __launch_bounds__(128) void k_stream_runs(const uint4 *in, size_t nchunks, uint32_t R, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const size_t run = (size_t)blockIdx.x * 2 + (threadIdx.x >> 6);
    uint32_t acc = 0;
    size_t c0 = run * R, c1 = c0 + R < nchunks ? c0 + R : nchunks;
    for (size_t c = c0; c < c1; c++) { const uint4 a = in[c * 64 + lane]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * 2));
    struct K { const char *name; void (*fn)(uint32_t *, uint32_t); int insts; };
    K ks[] = {{"v_add_u32", k_add, 8}, {"v_and_b32", k_and, 8}, {"v_pk_max_u16", k_pkmax, 8}, {"v_pk_min_u16", k_pkmin, 8},
              {"v_pk_sub_u16 clamp", k_pksubs, 8}, {"v_pk_add_u16", k_pkadd, 8}, {"v_pk_mul_lo_u16", k_pkmul, 8},
              {"v_pk_mad_u16", k_pkmad, 8}, {"v_perm_b32", k_perm, 8}, {"v_alignbit_b32", k_align, 8},
              {"v_max3_u32", k_max3, 8}, {"v_max_u16", k_maxu16, 8}, {"v_max3_u16", k_max3u16, 8}, {"v_fma_f32", k_fma, 8},
              {"v_pk_fma_f16", k_pkfma16, 8}, {"v_mad_u32_u24", k_madu24, 8}, {"v_dot4_u32_u8", k_dot4, 8},
              {"v_lshl_or_b32", k_lshlor, 8}, {"v_sad_u8", k_sadu8, 8}, {"v_cmp+v_addc", k_cmp_addc, 16},
              {"v_mov_dpp row_shr", k_movdpp, 8}, {"v_sub_u32_sdwa", k_subsdwa, 8},
              {"v_cndmask e64 sgpr", k_cnd64, 8}, {"v_cndmask vcc (no RAW)", k_cndnew, 8}, {"ds_bpermute+wait", k_bperm, 8},
              {"ds_swizzle+wait", k_swiz, 8}, {"v_add_dpp quad_perm", k_dppxor, 8}, {"v_add_dpp row_ror", k_dppror, 8},
              {"readlane+add", k_readlane, 16},
              {"v_cmp_gt_u32 vcc", k_cmpvcc, 8}, {"v_cmp_gt_u32 sgpr", k_cmpsg, 8}, {"v_cmp_gt_u16 vcc", k_cmpu16, 8},
              {"v_addc_co_u32 vcc", k_addc, 8}, {"v_addc_co_u32 sgpr", k_addcsg, 8}, {"v_max_u32", k_maxu32, 8},
              {"v_min_u32", k_minu32, 8}, {"v_sub_u32", k_subu32, 8}, {"v_lshrrev_b32", k_lshr, 8}, {"v_or_b32", k_or, 8},
              {"v_cndmask_b32", k_cndmask, 8}, {"v_max_f32", k_maxf32, 8}, {"v_sub_f32 clamp", k_subf32c, 8},
              {"v_mul_f32", k_mulf32, 8}, {"v_mac_f32", k_macf32, 8}, {"v_fma_f32 clamp", k_fmac, 8},
              {"v_cvt_f32_ubyte1", k_cvtub, 8}, {"v_cvt_f32_u32", k_cvtu32, 8}, {"v_mul_u32_u24", k_mulu24, 8},
              {"v_bfi_b32", k_bfi, 8}, {"v_bfe_u32", k_bfe, 8}, {"v_add3_u32", k_add3, 8}, {"v_lshl_add_u32", k_lshladd, 8},
              {"v_and_or_b32", k_andor, 8}, {"v_sub_u16", k_subu16, 8}, {"v_mul_lo_u16", k_mullou16, 8},
              {"v_mad_u16", k_madu16, 8}, {"v_xor_b32", k_xor, 8}, {"v_max_i32", k_maxi32, 8},
              {"v_dot4_i32_i8", k_dot4i, 8}, {"v_dot2_u32_u16", k_dot2, 8},
              {"[v_max_u32 + 1 salu] per v", k_v_s1, 8}, {"[v_max_u32 + 2 salu] per v", k_v_s2, 8},
              {"[v_cmp sgpr + 1 salu] per v", k_cmpsg_s1, 8}, {"[cmp,max,addc] per 3", k_cmp_addc2, 24},
              {"lds ring w128+3r128", k_lds_ring, 1}, {"4x (dpp wave_sh + add)", k_dpp_shift, 8}};
    for (int wps : {4}) {          // waves per SIMD
        printf("-- %d wave(s) per SIMD (grid %d x 256)\n", wps, cus * wps);
        for (auto &k : ks) {
            const float ms = time_ms([&] { hipLaunchKernelGGL(k.fn, dim3(cus * wps), dim3(256), 0, 0, out, 1u); }, 5);
            // per SIMD: wps waves x ITER x insts wave-instructions in ms
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * ITER * k.insts);
            printf("   %-26s %8.3f ms   %6.2f cyc/wave-inst @2.4GHz\n", k.name, ms, cyc);
        }
    }
    // streaming read ceilings
    const size_t bytes = 1ull << 30;
    uint4 *buf; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes));
    for (int g : {cus * 4, cus * 8, cus * 16, cus * 32}) {
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, buf, bytes / 16, out); }, 10);
        printf("stream grid-stride  grid %6d: %.3f ms  %.1f GB/s\n", g, ms, bytes / ms / 1e6);
    }
    for (uint32_t R : {16u, 32u, 64u, 128u, 256u}) {
        const size_t nchunks = bytes / 1024, nruns = (nchunks + R - 1) / R;
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_stream_runs, dim3((nruns + 1) / 2), dim3(128), 0, 0, buf, nchunks, R, out); }, 10);
        printf("stream wave-runs    R %4u (%zu runs): %.3f ms  %.1f GB/s\n", R, nruns, ms, bytes / ms / 1e6);
    }
    return 0;
}
