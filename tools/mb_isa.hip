// Micro-benchmark behind the round-3 threshold kernel: semantics and issue rates of the byte-SAD family and friends on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_isa.hip -o tools/mb_isa && ./tools/mb_isa
// Part 1 checks v_mqsad_pk_u16_u8 / v_qsad_pk_u16_u8 / v_dot4_i32_i8 against a host model (mask on zero REFERENCE bytes, 16-bit
// accumulators that wrap). Part 2 times dependent chains and 8 independent chains of each instruction with s_memtime, at 1 and 4
// waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void sem_kernel(const uint64_t* a, const uint32_t* ref, const uint64_t* acc, uint64_t* mq, uint64_t* q, int* d4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    mq[i] = __builtin_amdgcn_mqsad_pk_u16_u8(a[i], ref[i], acc[i]);
    q[i] = __builtin_amdgcn_qsad_pk_u16_u8(a[i], ref[i], acc[i]);
    d4[i] = __builtin_amdgcn_sdot4((int)(uint32_t)a[i], (int)ref[i], (int)(uint32_t)acc[i], false);
}

static uint32_t absd(uint32_t x, uint32_t y) { return x > y ? x - y : y - x; }
static uint64_t model(uint64_t a, uint32_t ref, uint64_t acc, bool masked) {
    uint64_t out = 0;
    for (int w = 0; w < 4; w++) {
        uint32_t s = (uint32_t)((acc >> (16 * w)) & 0xFFFF);
        for (int b = 0; b < 4; b++) {
            const uint32_t rb = (ref >> (8 * b)) & 0xFF, sb = (uint32_t)((a >> (8 * (w + b))) & 0xFF);
            if (masked && rb == 0) continue;
            s += absd(sb, rb);
        }
        out |= (uint64_t)(s & 0xFFFF) << (16 * w);
    }
    return out;
}

// Issue rates through inline assembly (the compiler folds chains of plain C arithmetic): every op works on 8 independent registers,
// dst = op(dst, r [, dst]).
#define RATE_OPS(X)                                                                   \
    X(0, "v_add_u32 %0, %0, %1", "v_add_u32")                                         \
    X(1, "v_and_b32 %0, %1, %0", "v_and_b32")                                         \
    X(2, "v_pk_add_u16 %0, %0, %1", "v_pk_add_u16")                                   \
    X(3, "v_pk_sub_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_sub_u16 op_sel") \
    X(4, "v_pk_lshrrev_b16 %0, 8, %0", "v_pk_lshrrev_b16")                            \
    X(5, "v_pk_mad_u16 %0, %0, %1, %0", "v_pk_mad_u16")                               \
    X(6, "v_perm_b32 %0, %0, %1, %0", "v_perm_b32")                                   \
    X(7, "v_alignbyte_b32 %0, %0, %1, 2", "v_alignbyte_b32")                          \
    X(8, "v_add3_u32 %0, %0, %1, %0", "v_add3_u32")                                   \
    X(9, "v_bfi_b32 %0, %1, %0, %0", "v_bfi_b32")                                     \
    X(10, "v_and_or_b32 %0, %0, %1, %0", "v_and_or_b32")                              \
    X(11, "v_dot4c_i32_i8 %0, %1, %0", "v_dot4c_i32_i8")                              \
    X(12, "v_sad_u8 %0, %0, %1, %0", "v_sad_u8")                                      \
    X(13, "v_sad_u16 %0, %0, %1, %0", "v_sad_u16")                                    \
    X(14, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp wave_shr") \
    X(15, "v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32")                           \
    X(16, "v_lshl_or_b32 %0, %0, 16, %1", "v_lshl_or_b32")                            \
    X(17, "v_lshrrev_b32 %0, 1, %0", "v_lshrrev_b32")                                 \
    X(18, "v_dot2_u32_u16 %0, %0, %1, %0", "v_dot2_u32_u16")                          \
    X(19, "v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp wave_shr") \
    X(20, "v_pk_mul_lo_u16 %0, %0, %1", "v_pk_mul_lo_u16")                            \
    X(21, "v_mad_u32_u24 %0, %0, %1, %0", "v_mad_u32_u24")                            \
    X(22, "v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2", "v_add_u16_sdwa") \
    X(23, "v_pk_ashrrev_i16 %0, 15, %0", "v_pk_ashrrev_i16")

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
    uint32_t w[8];
    for (int c = 0; c < 8; c++) w[c] = (seed + threadIdx.x * 977 + c * 131) * 0x9E3779B9u;
    const uint32_t r = seed | 0x01010101u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
#define X(id, text, name) if (OP == id) asm volatile(text : "+v"(w[c]) : "v"(r));
                RATE_OPS(X)
#undef X
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int c = 0; c < 8; c++) acc += w[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

// 64-bit forms through the builtins (their chains cannot be folded)
template <int OP>
__global__ __launch_bounds__(256) void rate64_kernel(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
    uint64_t v[8];
    for (int c = 0; c < 8; c++) v[c] = (uint64_t)(seed + threadIdx.x * 977 + c * 131) * 0x9E3779B97F4A7C15ull;
    const uint32_t r = seed | 0x01010101u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (OP == 0) v[c] = __builtin_amdgcn_mqsad_pk_u16_u8(v[c], r, v[c]);
                if (OP == 1) v[c] = __builtin_amdgcn_qsad_pk_u16_u8(v[c], r, v[c]);
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int c = 0; c < 8; c++) acc += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <class K>
static void run_rate(K kernel, const char* name, int waves_per_simd) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    uint64_t *out, *cyc;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&cyc, (size_t)blocks * 4 * 8));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 12345u);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 12345u);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto c : h) s += (double)c;
    const double ticks = s / h.size();                       // s_memtime ticks of one wave's loop (constant 100 MHz clock)
    const double ninst = (double)iters * 64;
    const double winst = (double)blocks * 4 * ninst;
    // cycles per instruction per SIMD from the wall time, assuming every SIMD holds `waves_per_simd` waves the whole time
    printf("  %-28s waves/SIMD %d : %7.1f G wave-instr/s chip-wide = %5.2f ns per instr per SIMD; one wave's loop: %.0f ticks of s_memtime\n", name, waves_per_simd,
           winst / (ms * 1e-3) * 1e-9, ms * 1e6 / (ninst * waves_per_simd), ticks);
    CK(hipFree(out));
    CK(hipFree(cyc));
}

int main() {
    const int n = 1 << 16;
    std::vector<uint64_t> a(n), acc(n), mq(n), q(n);
    std::vector<uint32_t> ref(n);
    std::vector<int> d4(n);
    srand(7);
    auto r64 = []() { return ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand(); };
    for (int i = 0; i < n; i++) {
        a[i] = r64(), acc[i] = (i & 1) ? r64() : 0xFFF0FFF8FFFFFFFEull;
        uint32_t rr = (uint32_t)r64();
        if (i % 4 == 0) rr = 0xFFFFFFFFu;
        if (i % 4 == 1) rr = 0xFFFFFF00u;
        if (i % 4 == 2) rr &= ((rand() & 1) ? 0xFFFFFFFFu : 0x00FFFF00u);
        if (i % 16 == 3) rr = 0;
        ref[i] = rr;
    }
    uint64_t *da, *dacc, *dmq, *dq;
    uint32_t* dref;
    int* dd4;
    CK(hipMalloc(&da, n * 8));
    CK(hipMalloc(&dacc, n * 8));
    CK(hipMalloc(&dmq, n * 8));
    CK(hipMalloc(&dq, n * 8));
    CK(hipMalloc(&dref, n * 4));
    CK(hipMalloc(&dd4, n * 4));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dacc, acc.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dref, ref.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sem_kernel, dim3(n / 256), dim3(256), 0, 0, da, dref, dacc, dmq, dq, dd4);
    CK(hipMemcpy(mq.data(), dmq, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(d4.data(), dd4, n * 4, hipMemcpyDeviceToHost));
    long bad_mq = 0, bad_q = 0, bad_d = 0, bad_mq_alt = 0;
    for (int i = 0; i < n; i++) {
        if (mq[i] != model(a[i], ref[i], acc[i], true)) {
            if (bad_mq < 4) printf("mqsad mismatch a=%016llx ref=%08x acc=%016llx got=%016llx want=%016llx\n", (unsigned long long)a[i], ref[i], (unsigned long long)acc[i],
                                   (unsigned long long)mq[i], (unsigned long long)model(a[i], ref[i], acc[i], true));
            bad_mq++;
        }
        if (q[i] != model(a[i], ref[i], acc[i], false)) bad_q++;
        int want = (int)(uint32_t)acc[i];
        for (int b = 0; b < 4; b++) want += (int)(int8_t)(a[i] >> (8 * b)) * (int)(int8_t)(ref[i] >> (8 * b));
        if (d4[i] != want) bad_d++;
    }
    printf("semantics: mqsad (mask = zero reference byte, wrapping u16 accumulators) mismatches %ld / %d; qsad %ld; dot4_i32_i8 %ld\n", bad_mq, n, bad_q, bad_d);
    (void)bad_mq_alt;
    printf("issue rates (256 CUs x 4 SIMDs, 8 independent registers per wave, inline assembly):\n");
#define X(id, text, name) run_rate(rate_kernel<id>, name, 1); run_rate(rate_kernel<id>, name, 4);
    RATE_OPS(X)
#undef X
    run_rate(rate64_kernel<0>, "v_mqsad_pk_u16_u8", 1);
    run_rate(rate64_kernel<0>, "v_mqsad_pk_u16_u8", 4);
    run_rate(rate64_kernel<1>, "v_qsad_pk_u16_u8", 4);
    return (bad_mq || bad_q || bad_d) ? 2 : 0;
}
