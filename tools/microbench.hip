// Issue-rate and vector-L1 look-up microbenchmarks behind the roofline labels of bench.py ("what's weak" #2):
//   part A  vector-instruction issue cost per SIMD at 1 / 2 / 6 waves per SIMD for the instruction kinds a node visit is made of
//   part B  divergent gathers of 64- and 48-byte records (4 / 3 / 2 / 1 x global_load_dwordx4 per lane and record) from tables that
//           sit in L1 / L2 / Infinity Cache: lane look-ups per clock per CU — the resource the trace kernels are bound by
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench      Run: tools/microbench > profiles/r03/microbench.jsonl
// Not product code; prints one JSON object per line.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// ---------------------------------------------------------------- part A: issue cost
// every kernel: `iters` iterations of 32 independent instructions of one kind on 8 (pairs of) registers; `out` keeps the compiler honest.
// One asm statement per 8 instructions so that the compiler inserts nothing between them.
#define ASM8_1(op, fmt)  asm volatile(op fmt(0) "\n" op fmt(1) "\n" op fmt(2) "\n" op fmt(3) "\n" op fmt(4) "\n" op fmt(5) "\n" op fmt(6) "\n" op fmt(7) \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(u) : "vcc")
#define ASM8_D(op, fmt)  asm volatile(op fmt(0) "\n" op fmt(1) "\n" op fmt(2) "\n" op fmt(3) "\n" op fmt(0) "\n" op fmt(1) "\n" op fmt(2) "\n" op fmt(3) \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db) : "vcc")
#define F_DAB(k)   " %" #k ", %" #k ", %8, %9"        /* dst, dst, a, b */
#define F_DA(k)    " %" #k ", %" #k ", %8"            /* dst, dst, a */
#define F_DU(k)    " %" #k ", %" #k ", %10"           /* dst, dst, u */
#define F_UD(k)    " %" #k ", %10, %" #k              /* dst, u, dst   (shift amount first) */
#define F_D1(k)    " %" #k ", %10"                    /* dst, u  (unary from u) */
#define F_DUA(k)   " %" #k ", %" #k ", %10, %8"       /* dst, dst, u, a */
#define F_CND(k)   " %" #k ", %" #k ", %8, vcc"
#define F_CMP(k)   " vcc, %" #k ", %8"
#define F_MIX(k)   " %" #k ", %10, %8, %9 op_sel_hi:[1,0,0]"      /* src0 = f16 (low half of u), src1, src2 = f32 */
#define F_SDWA(k)  " %" #k ", %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1"
#define FD_DAB(k)  " %" #k ", %" #k ", %4, %5"
#define FD_DA(k)   " %" #k ", %" #k ", %4"
#define KINDS(X) \
    X(0, "v_fma_f32", ASM8_1, F_DAB) X(1, "v_mul_f32", ASM8_1, F_DA) X(2, "v_add_f32", ASM8_1, F_DA) X(3, "v_sub_f32", ASM8_1, F_DA) \
    X(4, "v_max_f32", ASM8_1, F_DA) X(5, "v_min_f32", ASM8_1, F_DA) X(6, "v_max3_f32", ASM8_1, F_DAB) X(7, "v_med3_f32", ASM8_1, F_DAB) \
    X(8, "v_cvt_f32_ubyte1", ASM8_1, F_D1) X(9, "v_cvt_f32_u32", ASM8_1, F_D1) X(10, "v_cvt_f32_u32_sdwa", ASM8_1, F_SDWA) X(11, "v_fma_mix_f32", ASM8_1, F_MIX) \
    X(12, "v_cndmask_b32", ASM8_1, F_CND) X(13, "v_cmp_le_f32", ASM8_1, F_CMP) X(14, "v_min_u32", ASM8_1, F_DU) X(15, "v_max_u32", ASM8_1, F_DU) \
    X(16, "v_and_b32", ASM8_1, F_DU) X(17, "v_or_b32", ASM8_1, F_DU) X(18, "v_add_u32", ASM8_1, F_DU) X(19, "v_lshlrev_b32", ASM8_1, F_UD) \
    X(20, "v_perm_b32", ASM8_1, F_DUA) X(21, "v_bfe_u32", ASM8_1, F_DUA) X(22, "v_and_or_b32", ASM8_1, F_DUA) X(23, "v_lshl_add_u32", ASM8_1, F_DUA) \
    X(24, "v_ldexp_f32", ASM8_1, F_DU) X(25, "v_mov_b32", ASM8_1, F_D1) X(26, "v_mad_u32_u24", ASM8_1, F_DUA) X(27, "v_bfi_b32", ASM8_1, F_DUA) \
    X(28, "v_pk_fma_f32", ASM8_D, FD_DAB) X(29, "v_pk_mul_f32", ASM8_D, FD_DA) X(30, "v_min_f64", ASM8_D, FD_DA) X(31, "v_max_f64", ASM8_D, FD_DA) \
    X(32, "v_fma_f64", ASM8_D, FD_DAB) X(33, "v_add_f64", ASM8_D, FD_DA) X(34, "v_pk_add_f32", ASM8_D, FD_DA) X(35, "v_fmac_f32", ASM8_1, F_DA_FMAC)
#define F_DA_FMAC(k) " %" #k ", %8, %9"
constexpr int kKinds = 36 + 4;      // + ds_read_b32, ds_write_b32, ds_read_b128, (cmp + cndmask pair)
template <int KIND>
__global__ __launch_bounds__(256) void k_issue(int iters, float* out, unsigned long long* cycles) {
    float r0 = threadIdx.x * 1.0f, r1 = r0 + 1.0f, r2 = r0 + 2.0f, r3 = r0 + 3.0f, r4 = r0 + 4.0f, r5 = r0 + 5.0f, r6 = r0 + 6.0f, r7 = r0 + 7.0f;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3; const double da = 1.0001, db = 0.5;
    float a = 1.0001f, b = 0.5f;
    uint32_t u = (threadIdx.x * 2654435761u) & 0x03030303u;
    __shared__ float4 s_mem[256];
    s_mem[threadIdx.x] = make_float4(r0, r1, r2, r3);
    __syncthreads();
    const uint32_t ldsAddr = threadIdx.x * 4u, ldsAddr16 = threadIdx.x * 16u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define X(id, name, M, F) if (KIND == id) { M(name, F); M(name, F); M(name, F); M(name, F); }
        KINDS(X)
#undef X
        if (KIND == 36) {
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n"
                         "ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n"
                         "ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n"
                         "ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(ldsAddr) : "memory");
        } else if (KIND == 37) {
            asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1\n ds_write_b32 %8, %2\n ds_write_b32 %8, %3\n ds_write_b32 %8, %4\n ds_write_b32 %8, %5\n ds_write_b32 %8, %6\n ds_write_b32 %8, %7\n"
                         "ds_write_b32 %8, %0\n ds_write_b32 %8, %1\n ds_write_b32 %8, %2\n ds_write_b32 %8, %3\n ds_write_b32 %8, %4\n ds_write_b32 %8, %5\n ds_write_b32 %8, %6\n ds_write_b32 %8, %7\n"
                         "ds_write_b32 %8, %0\n ds_write_b32 %8, %1\n ds_write_b32 %8, %2\n ds_write_b32 %8, %3\n ds_write_b32 %8, %4\n ds_write_b32 %8, %5\n ds_write_b32 %8, %6\n ds_write_b32 %8, %7\n"
                         "ds_write_b32 %8, %0\n ds_write_b32 %8, %1\n ds_write_b32 %8, %2\n ds_write_b32 %8, %3\n ds_write_b32 %8, %4\n ds_write_b32 %8, %5\n ds_write_b32 %8, %6\n ds_write_b32 %8, %7\n s_waitcnt lgkmcnt(0)"
                         : : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(ldsAddr) : "memory");
        } else if (KIND == 38) {    // 8 x ds_read_b128 (counted as 8 instructions per statement x 4)
            float4 q0, q1, q2, q3;
            for (int k = 0; k < 8; ++k) {
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n s_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(ldsAddr16) : "memory");
                r0 += q0.x + q1.y + q2.z + q3.w;
            }
        } else if (KIND == 39) {    // dependent pair: v_cmp_le_f32 -> v_cndmask_b32 on the same register
            asm volatile("v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_le_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_le_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_le_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         "v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_le_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_le_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_le_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_le_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) + s_mem[(threadIdx.x + 1) & 255].x + (float)((d0 + d1) + (d2 + d3));
    if ((threadIdx.x & 63u) == 0u) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* kind_name(int k) {
#define X(id, name, M, F) if (k == id) return name;
    KINDS(X)
#undef X
    return k == 36 ? "ds_read_b32" : k == 37 ? "ds_write_b32" : k == 38 ? "ds_read_b128" : "v_cmp_le_f32+v_cndmask_b32 (dependent pair, per instruction)";
}

template <int KIND>
static void run_issue(int numCUs, int wavesPerSimd, float* dOut, unsigned long long* dCyc) {
    const int iters = 1500, blocks = numCUs * wavesPerSimd;      // 256-thread block = one wave on each SIMD of a CU
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, 50, dOut, dCyc);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, iters, dOut, dCyc);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    CHK(hipMemcpy(cyc.data(), dCyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto c : cyc) sum += (double)c;
    const double ticks = sum / (double)cyc.size();
    const double instrPerWave = (double)iters * 32.0;
    // s_memtime ticks are 100 MHz reference ticks here (ticks x 10 ns ~ the kernel's duration); the per-instruction cost is taken from the wall time
    std::printf("{\"bench\": \"issue\", \"instr\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"ns_per_instr_per_simd\": %.4f, \"cycles_at_2400MHz\": %.3f, \"memtime_ticks_per_wave\": %.0f}\n",
                kind_name(KIND), wavesPerSimd, ms, ms * 1e6 / (instrPerWave * wavesPerSimd), ms * 1e6 / (instrPerWave * wavesPerSimd) * 2.4, ticks);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}
template <int K> struct IssueAll { static void run(int cus, int w, float* o, unsigned long long* c) { IssueAll<K - 1>::run(cus, w, o, c); run_issue<K>(cus, w, o, c); } };
template <> struct IssueAll<-1> { static void run(int, int, float*, unsigned long long*) {} };

// ---------------------------------------------------------------- part B: divergent record gathers
// every lane walks its own pseudo-random sequence of records; LOADS x 16 bytes of each record are read (dwordx4 each)
template <int RECBYTES, int LOADS, int MODE>      // MODE 0: every lane its own record; 1: the 64 lanes of a wave share one record; 2: lane pairs share
__global__ __launch_bounds__(256) void k_gather(const float4* table, uint32_t nRecords, int iters, float* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    if (MODE == 1) s = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    if (MODE == 2) s = (blockIdx.x * 128u + (threadIdx.x >> 1)) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t rec = (uint32_t)(((unsigned long long)(s >> 4) * nRecords) >> 28);
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + (size_t)rec * RECBYTES);
        float4 v0 = p[0], v1, v2, v3;
        acc += (v0.x + v0.y) + (v0.z + v0.w);
        if (LOADS > 1) { v1 = p[1]; acc += (v1.x + v1.y) + (v1.z + v1.w); }
        if (LOADS > 2) { v2 = p[2]; acc += (v2.x + v2.y) + (v2.z + v2.w); }
        if (LOADS > 3) { v3 = p[3]; acc += (v3.x + v3.y) + (v3.z + v3.w); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int RECBYTES, int LOADS, int MODE>
static void run_gather(int numCUs, int wavesPerSimd, const float4* table, size_t tableBytes, float* dOut) {
    const int iters = 400, blocks = numCUs * wavesPerSimd;
    const uint32_t nRec = (uint32_t)(tableBytes / RECBYTES);
    hipLaunchKernelGGL((k_gather<RECBYTES, LOADS, MODE>), dim3(blocks), dim3(256), 0, 0, table, nRec, 50, dOut);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<RECBYTES, LOADS, MODE>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, dOut);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double laneLoads = (double)blocks * 256.0 * iters * LOADS;
    std::printf("{\"bench\": \"gather\", \"record_bytes\": %d, \"loads_per_record\": %d, \"sharing\": \"%s\", \"table_bytes\": %zu, \"waves_per_simd\": %d, \"kernel_ms\": %.4f, "
                "\"lane_loads_per_ns_per_cu\": %.4f, \"lane_loads_per_clk_per_cu_at_2400MHz\": %.4f, \"records_per_ns_chip\": %.3f, \"requested_TBps\": %.3f}\n",
                RECBYTES, LOADS, MODE == 0 ? "none" : MODE == 1 ? "wave" : "pair", tableBytes, wavesPerSimd, ms, laneLoads / (ms * 1e6) / numCUs, laneLoads / (ms * 1e6) / numCUs / 2.4,
                laneLoads / LOADS / (ms * 1e6), laneLoads * 16.0 / (ms * 1e-3) / 1e12);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

// ---------------------------------------------------------------- part C: the dependent load of a node visit
// A traversal's visits form a chain: the next node's address comes out of the node just loaded.  Every lane chases its own chain through a
// table of 64-byte records (all four quads loaded, the next index depends on all of them), the table filled with a pseudo-random
// successor per record; `extra` dependent multiply-adds between two loads stand for the visit's arithmetic.  ns per link at 1 wave per SIMD
// = the latency a lone wave pays per visit; at 6 waves per SIMD = what the chip delivers when latency is hidden.
__global__ void k_fill_chain(uint32_t* table, uint32_t nRecords) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    uint32_t s = i * 2654435761u + 97u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
    uint32_t* r = table + (size_t)i * 16;
    const uint32_t next = (uint32_t)(((unsigned long long)s * nRecords) >> 32);
    r[0] = next & 0xFFFFu; r[4] = next >> 16; r[8] = 0u; r[12] = 0u;             // the successor is spread over the quads: all four loads are on the chain
    for (int k = 0; k < 16; ++k) if (k != 0 && k != 4 && k != 8 && k != 12) r[k] = 0u;
}
template <int EXTRA>
__global__ __launch_bounds__(256) void k_chase(const uint32_t* table, uint32_t nRecords, int iters, uint32_t* out) {
    uint32_t idx = (uint32_t)(((unsigned long long)((blockIdx.x * 256u + threadIdx.x) * 2654435761u) * nRecords) >> 32);
    float f = 1.0f;
    for (int i = 0; i < iters; ++i) {
        const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * 16);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        idx = (a.x | (b.x << 16)) + c.x + d.x;
        if (EXTRA) {
#pragma unroll
            for (int k = 0; k < EXTRA; ++k) f = __builtin_fmaf(f, 1.0000001f, (float)(idx & 1u));
            idx += (f > 3.0e38f) ? 1u : 0u;                                            // never true: keeps the arithmetic on the chain
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = idx + (uint32_t)f;
}
template <int EXTRA>
static void run_chase(int numCUs, int wavesPerSimd, uint32_t* table, size_t tableBytes, uint32_t* dOut) {
    const int iters = 300, blocks = numCUs * wavesPerSimd;
    const uint32_t nRec = (uint32_t)(tableBytes / 64);
    hipLaunchKernelGGL(k_fill_chain, dim3((nRec + 255) / 256), dim3(256), 0, 0, table, nRec);
    hipLaunchKernelGGL((k_chase<EXTRA>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)table, nRec, 30, dOut);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chase<EXTRA>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)table, nRec, iters, dOut);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("{\"bench\": \"chase\", \"table_bytes\": %zu, \"waves_per_simd\": %d, \"dependent_fma_per_link\": %d, \"kernel_ms\": %.4f, \"ns_per_link\": %.1f, "
                "\"lane_links_per_ns_chip\": %.2f}\n", tableBytes, wavesPerSimd, EXTRA, ms, ms * 1e6 / iters, (double)blocks * 256.0 * iters / (ms * 1e6));
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const bool onlyChase = argc > 1 && std::strcmp(argv[1], "chase") == 0;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int numCUs = prop.multiProcessorCount;
    std::printf("{\"bench\": \"device\", \"name\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, numCUs, prop.clockRate);
    float* dOut; unsigned long long* dCyc;
    CHK(hipMalloc((void**)&dOut, (size_t)numCUs * 8 * 256 * 4)); CHK(hipMalloc((void**)&dCyc, (size_t)numCUs * 8 * 4 * 8));
    {
        uint32_t* chain; CHK(hipMalloc((void**)&chain, (size_t)1 << 30));
        for (size_t bytes : {(size_t)16 << 10, (size_t)1 << 20, (size_t)12 << 20, (size_t)64 << 20, (size_t)1 << 30})
            for (int w : {1, 2, 6}) { run_chase<0>(numCUs, w, chain, bytes, (uint32_t*)dOut); run_chase<64>(numCUs, w, chain, bytes, (uint32_t*)dOut); }
        CHK(hipFree(chain));
    }
    if (onlyChase) return 0;
    for (int w : {1, 2, 6}) IssueAll<kKinds - 1>::run(numCUs, w, dOut, dCyc);
    const size_t big = 64u << 20;
    float4* table; CHK(hipMalloc((void**)&table, big)); CHK(hipMemset(table, 0, big));
    for (size_t bytes : {(size_t)16 << 10, (size_t)2 << 20, (size_t)12 << 20, (size_t)60 << 20}) {
        run_gather<64, 4, 0>(numCUs, 6, table, bytes, dOut); run_gather<64, 3, 0>(numCUs, 6, table, bytes, dOut); run_gather<64, 2, 0>(numCUs, 6, table, bytes, dOut);
        run_gather<64, 1, 0>(numCUs, 6, table, bytes, dOut); run_gather<48, 3, 0>(numCUs, 6, table, bytes, dOut);
        run_gather<64, 4, 1>(numCUs, 6, table, bytes, dOut); run_gather<64, 4, 2>(numCUs, 6, table, bytes, dOut);
    }
    run_gather<64, 4, 0>(numCUs, 4, table, (size_t)12 << 20, dOut); run_gather<48, 3, 0>(numCUs, 4, table, (size_t)12 << 20, dOut);
    run_gather<64, 4, 0>(numCUs, 8, table, (size_t)12 << 20, dOut); run_gather<48, 3, 0>(numCUs, 8, table, (size_t)12 << 20, dOut);
    return 0;
}
