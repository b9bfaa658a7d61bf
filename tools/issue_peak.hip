// issue_peak.hip -- microbenchmark behind the bench line's `roofline` peaks.
//
// Measures, on the GPU it runs on, how many shader cycles one SIMD needs per wave64 instruction
// for the instruction classes the pooled render kernel is made of, at 1 / 2 / 4 / 8 waves per SIMD:
// plain and packed fp32 VALU, v_cndmask with an SGPR mask, compares that write SGPR pairs, the
// IEEE divide / sqrt expansion pieces, mbcnt, SALU mask logic, VALU+SALU mixes, and the LDS
// operations with the kernel's own address patterns (ds_bpermute pulls, 64-byte node records
// at random indices as AoS and as 4 planes, appends with a shared dump slot, per-slot atomics).
//
//   hipcc --offload-arch=gfx950 -O3 -o build/issue_peak tools/issue_peak.hip && build/issue_peak
//
// Output: one line per (op, waves/SIMD): shader cycles per wave-instruction per SIMD (s_memtime
// ticks), per CU for LDS operations, and the chip-wide rate in G wave-instr/s at 2.4 GHz.
// Design tool + evidence (profiles/r02/issue_peak.txt); not a product path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

struct Args {
  unsigned long long *cyc;   // per wave: cycles of the timed loop
  int *sink;
  int iters;
  int pattern;               // LDS address pattern selector
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// LDS address patterns (byte addresses).  The LDS image is 64 KB of small numbers.
//  0: lane-linear dwords (conflict free)            a_k = 16*lane (+ 1024*k)   [b128-safe]
//  1: all lanes the same address
//  2: random lane selector for ds_bpermute          a_k = 4*rand(64)
//  3: random 64-byte record, AoS: a0..a3 = the record's four 16-byte quarters (a4..a7 a second record)
//  4: random record, 4 planes (SoA of 16-byte quarters): a_k = plane_k + 16*rec
//  5: half of the lanes (random) -> one shared dump address, the rest consecutive dwords
//  6: half of the lanes -> per-lane dump address (dump + 4*lane), the rest consecutive dwords
//  7: per-slot counters: 4*slot with runs (neighbouring lanes often share a slot)
//  8: ds_bpermute selectors with runs
// 11: ray-table planes (256 B per component), index = slot with runs;  12: 8-byte stride linear;  13: 16-byte ray records
//  9: lane-linear dwords, 4-byte stride: a_k = 4*lane + 256*k
// 10: random record among 399, planes, but records come in sibling pairs (lane 2i, 2i+1 -> rec, rec+1)
__device__ __forceinline__ void make_addrs(int pattern, int lane, int wave, int &a0, int &a1, int &a2, int &a3, int &a4,
                                           int &a5, int &a6, int &a7) {
  const unsigned h = hash32(lane * 2654435761u + wave * 97u + 13u);
  switch (pattern) {
  default:
  case 0: a0 = 16 * lane; a1 = a0 + 1024; a2 = a0 + 2048; a3 = a0 + 3072; a4 = a0 + 4096; a5 = a0 + 5120; a6 = a0 + 6144; a7 = a0 + 7168; break;
  case 9: a0 = 4 * lane; a1 = a0 + 256; a2 = a0 + 512; a3 = a0 + 768; a4 = a0 + 1024; a5 = a0 + 1280; a6 = a0 + 1536; a7 = a0 + 1792; break;
  case 1: a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = 64; break;
  case 2: a0 = 4 * (h & 63); a1 = 4 * ((h >> 6) & 63); a2 = 4 * ((h >> 12) & 63); a3 = 4 * ((h >> 18) & 63); a4 = 4 * ((h >> 24) & 63);
          a5 = 4 * ((h * 7u >> 5) & 63); a6 = 4 * ((h * 13u >> 9) & 63); a7 = 4 * ((h * 29u >> 11) & 63); break;
  case 3: { const int rec = (int)(h % 399u); a0 = 64 * rec; a1 = a0 + 16; a2 = a0 + 32; a3 = a0 + 48;
            const int rec2 = (int)((h >> 11) % 399u); a4 = 64 * rec2; a5 = a4 + 16; a6 = a4 + 32; a7 = a4 + 48; } break;
  case 4: { const int rec = (int)(h % 399u); a0 = 16 * rec; a1 = a0 + 6400; a2 = a0 + 12800; a3 = a0 + 19200;
            const int rec2 = (int)((h >> 11) % 399u); a4 = 16 * rec2; a5 = a4 + 6400; a6 = a4 + 12800; a7 = a4 + 19200; } break;
  case 10: { const unsigned hp = hash32((lane >> 1) * 2654435761u + wave * 97u + 13u);
            const int rec = (int)(hp % 398u) + (lane & 1); a0 = 16 * rec; a1 = a0 + 6400; a2 = a0 + 12800; a3 = a0 + 19200;
            const int rec2 = (int)((hp >> 11) % 398u) + (lane & 1); a4 = 16 * rec2; a5 = a4 + 6400; a6 = a4 + 12800; a7 = a4 + 19200; } break;
  case 5: case 6: {
    const bool app = (h & 1u) != 0u;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(app);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    const int dump = pattern == 5 ? 32768 : 32768 + 4 * lane;
    a0 = app ? 4096 + 4 * rank : dump; a1 = app ? 8192 + 4 * rank : dump; a2 = app ? 12288 + 4 * rank : dump; a3 = app ? 16384 + 4 * rank : dump;
    a4 = app ? 20480 + 4 * rank : dump; a5 = app ? 24576 + 4 * rank : dump; a6 = app ? 28672 + 4 * rank : dump; a7 = app ? 36864 + 4 * rank : dump;
  } break;
  case 11: {   // ray table: one 256-byte plane per component, dword index = slot (runs)
    const int grp = lane >> (hash32(lane >> 3) & 3u);
    const int slot = (int)(hash32(grp * 977u + wave) & 63u);
    a0 = 4 * slot; a1 = a0 + 256; a2 = a0 + 512; a3 = a0 + 768; a4 = a0 + 1024; a5 = a0 + 1280; a6 = a0 + 1536; a7 = a0 + 1792;
  } break;
  case 12: a0 = 8 * lane; a1 = a0 + 512; a2 = a0 + 1024; a3 = a0 + 1536; a4 = a0 + 2048; a5 = a0 + 2560; a6 = a0 + 3072; a7 = a0 + 3584; break;
  case 13: {   // ray table as 16-byte records per slot (4 planes of 64 x 16 B), slot with runs
    const int grp = lane >> (hash32(lane >> 3) & 3u);
    const int slot = (int)(hash32(grp * 977u + wave) & 63u);
    a0 = 16 * slot; a1 = a0 + 1024; a2 = a0 + 2048; a3 = a0 + 3072; a4 = a0 + 4096; a5 = a0 + 5120; a6 = a0 + 6144; a7 = a0 + 7168;
  } break;
  case 7: case 8: {
    const int grp = lane >> (hash32(lane >> 3) & 3u);   // neighbouring lanes share a slot (runs of 1..8)
    const unsigned hg = hash32(grp * 977u + wave);
    a0 = 4 * (hg & 63); a1 = 4 * ((hg >> 6) & 63); a2 = 4 * ((hg >> 12) & 63); a3 = 4 * ((hg >> 18) & 63); a4 = 4 * ((hg >> 24) & 63);
    a5 = 4 * ((hg * 7u >> 5) & 63); a6 = 4 * ((hg * 13u >> 9) & 63); a7 = 4 * ((hg * 29u >> 11) & 63);
    if (pattern == 7) { a0 += 40960; a1 += 40960; a2 += 40960; a3 += 40960; a4 += 40960; a5 += 40960; a6 += 40960; a7 += 40960; }
  } break;
  }
}

// One kernel per op.  The timed loop is ONE asm statement: data registers v[20:51], masks
// s[60:67], LDS addresses in v[56:63] (copied from operands), loop counter s59.
#define KERNEL(NAME, REPT, NINST, BODY)                                                              \
  __global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void NAME(Args A) {                                             \
    extern __shared__ unsigned lds[];                                                                \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                      \
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = hash32(i) & 0x3ffu;               \
    __syncthreads();                                                                                 \
    int a0, a1, a2, a3, a4, a5, a6, a7;                                                              \
    make_addrs(A.pattern, lane, wave, a0, a1, a2, a3, a4, a5, a6, a7);                               \
    const float fl = lane * 0.5f + 1.0f;                                                             \
    unsigned long long t0, t1;                                                                       \
    const int iters = __builtin_amdgcn_readfirstlane(A.iters);                                       \
    __syncthreads();                                                                                 \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                  \
    asm volatile(                                                                                    \
        "v_mov_b32 v56, %3\nv_mov_b32 v57, %4\nv_mov_b32 v58, %5\nv_mov_b32 v59, %6\n"           \
        "v_mov_b32 v60, %7\nv_mov_b32 v61, %8\nv_mov_b32 v62, %9\nv_mov_b32 v63, %10\n"          \
        "v_mov_b32 v20, %11\nv_add_f32 v21, 1.0, v20\nv_add_f32 v22, 2.0, v20\n"                \
        "v_add_f32 v23, 1.0, v22\nv_add_f32 v24, 2.0, v22\nv_add_f32 v25, 1.0, v24\n"          \
        "v_add_f32 v26, 2.0, v24\nv_add_f32 v27, 1.0, v26\nv_add_f32 v28, 2.0, v26\n"          \
        "v_add_f32 v29, 1.0, v28\nv_add_f32 v30, 2.0, v28\nv_add_f32 v31, 1.0, v30\n"          \
        "v_add_f32 v32, 2.0, v30\nv_add_f32 v33, 1.0, v32\nv_add_f32 v34, 2.0, v32\n"          \
        "v_add_f32 v35, 1.0, v34\n"                                                                \
        "v_mov_b32 v36, 0x3f800001\nv_mov_b32 v37, 0x3f7fffff\nv_mov_b32 v38, 1\nv_mov_b32 v39, 0\n" \
        "v_mov_b32 v40, 0\nv_mov_b32 v41, 0\nv_mov_b32 v42, 0\nv_mov_b32 v43, 0\n"               \
        "v_mov_b32 v44, 0\nv_mov_b32 v45, 0\nv_mov_b32 v46, 0\nv_mov_b32 v47, 0\n"               \
        "v_mov_b32 v48, 0\nv_mov_b32 v49, 0\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0\n"               \
        "s_mov_b32 s60, 0x55555555\ns_mov_b32 s61, 0x33333333\ns_mov_b32 s62, 0x0f0f0f0f\n"          \
        "s_mov_b32 s63, 0x00ff00ff\ns_mov_b32 s64, 0x12345678\ns_mov_b32 s65, 0x9abcdef0\n"          \
        "s_mov_b32 s66, 0xdeadbeef\ns_mov_b32 s67, 0x0badf00d\n"                                     \
        "s_mov_b64 vcc, s[60:61]\n"                                                                  \
        "s_mov_b32 s59, %2\n"                                                                        \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                            \
        "s_memtime %0\n"                                                                             \
        "s_waitcnt lgkmcnt(0)\n"                                                                     \
        "1:\n"                                                                                       \
        ".rept " #REPT "\n" BODY "\n.endr\n"                                                         \
        "s_sub_u32 s59, s59, 1\n"                                                                    \
        "s_cmp_lg_u32 s59, 0\n"                                                                      \
        "s_cbranch_scc1 1b\n"                                                                        \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                            \
        "s_memtime %1\n"                                                                             \
        "s_waitcnt lgkmcnt(0)\n"                                                                     \
        : "=&s"(t0), "=&s"(t1)                                                                       \
        : "s"(iters), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(fl) \
        : "vcc", "scc", "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",    \
          "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",    \
          "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",    \
          "v50", "v51", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s60",     \
          "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s59");                     \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                  \
    if (lane == 0) {                                                                                 \
      A.cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;                                        \
      A.cyc[32768 + blockIdx.x * (blockDim.x >> 6) + wave] = r1 - r0;                                \
    }                                                                                                \
  }                                                                                                  \
  static const int NAME##_ninst = (REPT) * (NINST);

// ---- VALU: 8 independent registers unless noted ----
#define R8(OP, TAIL) OP " v20,v20" TAIL "\n" OP " v21,v21" TAIL "\n" OP " v22,v22" TAIL "\n" OP " v23,v23" TAIL "\n" \
                     OP " v24,v24" TAIL "\n" OP " v25,v25" TAIL "\n" OP " v26,v26" TAIL "\n" OP " v27,v27" TAIL
KERNEL(k_add, 16, 8, R8("v_add_f32", ",v36"))
KERNEL(k_mul, 16, 8, R8("v_mul_f32", ",v36"))
KERNEL(k_fma, 16, 8, R8("v_fma_f32", ",v36,v37"))
KERNEL(k_max3, 16, 8, R8("v_max3_f32", ",v36,v37"))
KERNEL(k_min, 16, 8, R8("v_min_f32", ",v36"))
KERNEL(k_add_dep, 16, 8, "v_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36\n"
                         "v_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v20,v20,v36")
KERNEL(k_add_dep2, 16, 8, "v_add_f32 v20,v20,v36\nv_add_f32 v21,v21,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v21,v21,v36\n"
                          "v_add_f32 v20,v20,v36\nv_add_f32 v21,v21,v36\nv_add_f32 v20,v20,v36\nv_add_f32 v21,v21,v36")
KERNEL(k_pk_mul, 16, 8, "v_pk_mul_f32 v[20:21],v[20:21],v[36:37]\nv_pk_mul_f32 v[22:23],v[22:23],v[36:37]\n"
                        "v_pk_mul_f32 v[24:25],v[24:25],v[36:37]\nv_pk_mul_f32 v[26:27],v[26:27],v[36:37]\n"
                        "v_pk_mul_f32 v[28:29],v[28:29],v[36:37]\nv_pk_mul_f32 v[30:31],v[30:31],v[36:37]\n"
                        "v_pk_mul_f32 v[32:33],v[32:33],v[36:37]\nv_pk_mul_f32 v[34:35],v[34:35],v[36:37]")
KERNEL(k_pk_add, 16, 8, "v_pk_add_f32 v[20:21],v[20:21],v[36:37]\nv_pk_add_f32 v[22:23],v[22:23],v[36:37]\n"
                        "v_pk_add_f32 v[24:25],v[24:25],v[36:37]\nv_pk_add_f32 v[26:27],v[26:27],v[36:37]\n"
                        "v_pk_add_f32 v[28:29],v[28:29],v[36:37]\nv_pk_add_f32 v[30:31],v[30:31],v[36:37]\n"
                        "v_pk_add_f32 v[32:33],v[32:33],v[36:37]\nv_pk_add_f32 v[34:35],v[34:35],v[36:37]")
KERNEL(k_cndmask_vcc, 16, 8, R8("v_cndmask_b32", ",v36,vcc"))
KERNEL(k_cndmask_sgpr, 16, 8, "v_cndmask_b32 v20,v20,v36,s[60:61]\nv_cndmask_b32 v21,v21,v36,s[62:63]\n"
                              "v_cndmask_b32 v22,v22,v36,s[64:65]\nv_cndmask_b32 v23,v23,v36,s[66:67]\n"
                              "v_cndmask_b32 v24,v24,v36,s[60:61]\nv_cndmask_b32 v25,v25,v36,s[62:63]\n"
                              "v_cndmask_b32 v26,v26,v36,s[64:65]\nv_cndmask_b32 v27,v27,v36,s[66:67]")
KERNEL(k_cndmask_vcc_e64, 16, 8, "v_cndmask_b32_e64 v20,v20,v36,vcc\nv_cndmask_b32_e64 v21,v21,v36,vcc\nv_cndmask_b32_e64 v22,v22,v36,vcc\nv_cndmask_b32_e64 v23,v23,v36,vcc\n"
                                 "v_cndmask_b32_e64 v24,v24,v36,vcc\nv_cndmask_b32_e64 v25,v25,v36,vcc\nv_cndmask_b32_e64 v26,v26,v36,vcc\nv_cndmask_b32_e64 v27,v27,v36,vcc")
KERNEL(k_cmp_cnd_vcc, 16, 8, "v_cmp_lt_f32 vcc,v28,v36\nv_cndmask_b32 v20,v20,v36,vcc\nv_cmp_lt_f32 vcc,v29,v36\nv_cndmask_b32 v21,v21,v36,vcc\n"
                             "v_cmp_lt_f32 vcc,v30,v36\nv_cndmask_b32 v22,v22,v36,vcc\nv_cmp_lt_f32 vcc,v31,v36\nv_cndmask_b32 v23,v23,v36,vcc")
KERNEL(k_cmp_cnd_sgpr, 16, 8, "v_cmp_lt_f32 s[60:61],v28,v36\nv_cndmask_b32 v20,v20,v36,s[60:61]\nv_cmp_lt_f32 s[62:63],v29,v36\nv_cndmask_b32 v21,v21,v36,s[62:63]\n"
                              "v_cmp_lt_f32 s[64:65],v30,v36\nv_cndmask_b32 v22,v22,v36,s[64:65]\nv_cmp_lt_f32 s[66:67],v31,v36\nv_cndmask_b32 v23,v23,v36,s[66:67]")
KERNEL(k_sub, 16, 8, R8("v_sub_f32", ",v36"))
KERNEL(k_bfe, 16, 8, R8("v_bfe_u32", ",2,6"))
KERNEL(k_and, 16, 8, R8("v_and_b32", ",v38"))
KERNEL(k_add_u32v, 16, 8, R8("v_add_u32", ",v38"))
KERNEL(k_cmp_vcc, 16, 8, "v_cmp_lt_f32 vcc,v20,v36\nv_cmp_lt_f32 vcc,v21,v36\nv_cmp_lt_f32 vcc,v22,v36\nv_cmp_lt_f32 vcc,v23,v36\n"
                         "v_cmp_lt_f32 vcc,v24,v36\nv_cmp_lt_f32 vcc,v25,v36\nv_cmp_lt_f32 vcc,v26,v36\nv_cmp_lt_f32 vcc,v27,v36")
KERNEL(k_cmp_sgpr, 16, 8, "v_cmp_lt_f32 s[60:61],v20,v36\nv_cmp_lt_f32 s[62:63],v21,v36\nv_cmp_lt_f32 s[64:65],v22,v36\nv_cmp_lt_f32 s[66:67],v23,v36\n"
                          "v_cmp_lt_f32 s[60:61],v24,v36\nv_cmp_lt_f32 s[62:63],v25,v36\nv_cmp_lt_f32 s[64:65],v26,v36\nv_cmp_lt_f32 s[66:67],v27,v36")
// compare -> scalar logic -> select chain, as in the BOX operation (v_cmp writes SGPRs, SALU combines, v_cndmask reads)
KERNEL(k_cmp_salu_cnd, 16, 6, "v_cmp_lt_f32 s[60:61],v20,v36\nv_cmp_lt_f32 s[62:63],v21,v36\ns_and_b64 s[64:65],s[60:61],s[62:63]\n"
                              "v_cndmask_b32 v22,v22,v36,s[64:65]\nv_add_f32 v23,v23,v36\nv_add_f32 v24,v24,v36")
KERNEL(k_rcp, 16, 8, R8("v_rcp_f32", ""))
KERNEL(k_sqrt, 16, 8, R8("v_sqrt_f32", ""))
KERNEL(k_div_scale, 16, 8, "v_div_scale_f32 v20,vcc,v20,v36,v20\nv_div_scale_f32 v21,vcc,v21,v36,v21\nv_div_scale_f32 v22,vcc,v22,v36,v22\n"
                           "v_div_scale_f32 v23,vcc,v23,v36,v23\nv_div_scale_f32 v24,vcc,v24,v36,v24\nv_div_scale_f32 v25,vcc,v25,v36,v25\n"
                           "v_div_scale_f32 v26,vcc,v26,v36,v26\nv_div_scale_f32 v27,vcc,v27,v36,v27")
KERNEL(k_div_fmas, 16, 8, R8("v_div_fmas_f32", ",v36,v37"))
KERNEL(k_div_fixup, 16, 8, R8("v_div_fixup_f32", ",v36,v37"))
KERNEL(k_mbcnt, 16, 8, "v_mbcnt_lo_u32_b32 v20,s60,0\nv_mbcnt_hi_u32_b32 v20,s61,v20\nv_mbcnt_lo_u32_b32 v21,s62,0\nv_mbcnt_hi_u32_b32 v21,s63,v21\n"
                       "v_mbcnt_lo_u32_b32 v22,s64,0\nv_mbcnt_hi_u32_b32 v22,s65,v22\nv_mbcnt_lo_u32_b32 v23,s66,0\nv_mbcnt_hi_u32_b32 v23,s67,v23")
KERNEL(k_lshl_add, 16, 8, R8("v_lshl_add_u32", ",2,v38"))
// round 5: candidates for the box test's sign swap and the append's address selects (is there a 2-cycle form of a select?)
KERNEL(k_bfi, 16, 8, R8("v_bfi_b32", ",v38,v36"))
KERNEL(k_xor, 16, 8, R8("v_xor_b32", ",v38"))
KERNEL(k_or, 16, 8, R8("v_or_b32", ",v38"))
KERNEL(k_lshlrev, 16, 8, "v_lshlrev_b32 v20,2,v20\nv_lshlrev_b32 v21,2,v21\nv_lshlrev_b32 v22,2,v22\nv_lshlrev_b32 v23,2,v23\n"
                         "v_lshlrev_b32 v24,2,v24\nv_lshlrev_b32 v25,2,v25\nv_lshlrev_b32 v26,2,v26\nv_lshlrev_b32 v27,2,v27")
KERNEL(k_ashrrev, 16, 8, "v_ashrrev_i32 v20,31,v20\nv_ashrrev_i32 v21,31,v21\nv_ashrrev_i32 v22,31,v22\nv_ashrrev_i32 v23,31,v23\n"
                         "v_ashrrev_i32 v24,31,v24\nv_ashrrev_i32 v25,31,v25\nv_ashrrev_i32 v26,31,v26\nv_ashrrev_i32 v27,31,v27")
KERNEL(k_max, 16, 8, R8("v_max_f32", ",v36"))
KERNEL(k_med3, 16, 8, R8("v_med3_f32", ",v36,v37"))
KERNEL(k_perm, 16, 8, R8("v_perm_b32", ",v36,v38"))
KERNEL(k_and_or, 16, 8, R8("v_and_or_b32", ",v38,v39"))
KERNEL(k_lshl_or, 16, 8, R8("v_lshl_or_b32", ",2,v38"))
KERNEL(k_add3, 16, 8, R8("v_add3_u32", ",v38,v39"))
KERNEL(k_sub_u32, 16, 8, R8("v_sub_u32", ",v38"))
KERNEL(k_min_u32, 16, 8, R8("v_min_u32", ",v38"))
KERNEL(k_fmac, 16, 8, "v_fmac_f32 v20,v36,v37\nv_fmac_f32 v21,v36,v37\nv_fmac_f32 v22,v36,v37\nv_fmac_f32 v23,v36,v37\n"
                      "v_fmac_f32 v24,v36,v37\nv_fmac_f32 v25,v36,v37\nv_fmac_f32 v26,v36,v37\nv_fmac_f32 v27,v36,v37")
KERNEL(k_pk_fma, 16, 8, "v_pk_fma_f32 v[20:21],v[20:21],v[36:37],v[36:37]\nv_pk_fma_f32 v[22:23],v[22:23],v[36:37],v[36:37]\n"
                        "v_pk_fma_f32 v[24:25],v[24:25],v[36:37],v[36:37]\nv_pk_fma_f32 v[26:27],v[26:27],v[36:37],v[36:37]\n"
                        "v_pk_fma_f32 v[28:29],v[28:29],v[36:37],v[36:37]\nv_pk_fma_f32 v[30:31],v[30:31],v[36:37],v[36:37]\n"
                        "v_pk_fma_f32 v[32:33],v[32:33],v[36:37],v[36:37]\nv_pk_fma_f32 v[34:35],v[34:35],v[36:37],v[36:37]")
KERNEL(k_pk_mov, 16, 8, "v_pk_mov_b32 v[20:21],v[36:37],v[36:37]\nv_pk_mov_b32 v[22:23],v[36:37],v[36:37]\n"
                        "v_pk_mov_b32 v[24:25],v[36:37],v[36:37]\nv_pk_mov_b32 v[26:27],v[36:37],v[36:37]\n"
                        "v_pk_mov_b32 v[28:29],v[36:37],v[36:37]\nv_pk_mov_b32 v[30:31],v[36:37],v[36:37]\n"
                        "v_pk_mov_b32 v[32:33],v[36:37],v[36:37]\nv_pk_mov_b32 v[34:35],v[36:37],v[36:37]")
KERNEL(k_mov, 16, 8, "v_mov_b32 v20,v36\nv_mov_b32 v21,v36\nv_mov_b32 v22,v36\nv_mov_b32 v23,v36\nv_mov_b32 v24,v36\nv_mov_b32 v25,v36\nv_mov_b32 v26,v36\nv_mov_b32 v27,v36")
KERNEL(k_readfirstlane, 16, 8, "v_readfirstlane_b32 s60,v20\nv_readfirstlane_b32 s61,v21\nv_readfirstlane_b32 s62,v22\nv_readfirstlane_b32 s63,v23\n"
                               "v_readfirstlane_b32 s64,v24\nv_readfirstlane_b32 s65,v25\nv_readfirstlane_b32 s66,v26\nv_readfirstlane_b32 s67,v27")
// ---- SALU ----
KERNEL(k_salu, 16, 8, "s_and_b64 s[60:61],s[60:61],s[62:63]\ns_or_b64 s[62:63],s[62:63],s[64:65]\ns_xor_b64 s[64:65],s[64:65],s[66:67]\ns_andn2_b64 s[66:67],s[66:67],s[60:61]\n"
                      "s_and_b64 s[60:61],s[60:61],s[62:63]\ns_or_b64 s[62:63],s[62:63],s[64:65]\ns_xor_b64 s[64:65],s[64:65],s[66:67]\ns_andn2_b64 s[66:67],s[66:67],s[60:61]")
KERNEL(k_bcnt, 16, 8, "s_bcnt1_i32_b64 s68,s[60:61]\ns_bcnt1_i32_b64 s69,s[62:63]\ns_bcnt1_i32_b64 s68,s[64:65]\ns_bcnt1_i32_b64 s69,s[66:67]\n"
                      "s_bcnt1_i32_b64 s68,s[60:61]\ns_bcnt1_i32_b64 s69,s[62:63]\ns_bcnt1_i32_b64 s68,s[64:65]\ns_bcnt1_i32_b64 s69,s[66:67]")
KERNEL(k_snop, 16, 8, "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0")
// ---- mixes (instruction counts: all instructions) ----
KERNEL(k_mix_v1s1, 16, 8, "v_add_f32 v20,v20,v36\ns_and_b64 s[60:61],s[60:61],s[62:63]\nv_add_f32 v21,v21,v36\ns_or_b64 s[62:63],s[62:63],s[64:65]\n"
                          "v_add_f32 v22,v22,v36\ns_xor_b64 s[64:65],s[64:65],s[66:67]\nv_add_f32 v23,v23,v36\ns_andn2_b64 s[66:67],s[66:67],s[60:61]")
KERNEL(k_mix_v2s1, 16, 9, "v_add_f32 v20,v20,v36\nv_add_f32 v21,v21,v36\ns_and_b64 s[60:61],s[60:61],s[62:63]\nv_add_f32 v22,v22,v36\nv_add_f32 v23,v23,v36\n"
                          "s_or_b64 s[62:63],s[62:63],s[64:65]\nv_add_f32 v24,v24,v36\nv_add_f32 v25,v25,v36\ns_xor_b64 s[64:65],s[64:65],s[66:67]")
// ---- LDS (results land in v36..v51; one lgkmcnt wait per 8) ----
KERNEL(k_bperm, 8, 8, "ds_bpermute_b32 v40,v56,v20\nds_bpermute_b32 v41,v57,v21\nds_bpermute_b32 v42,v58,v22\nds_bpermute_b32 v43,v59,v23\n"
                      "ds_bpermute_b32 v44,v60,v24\nds_bpermute_b32 v45,v61,v25\nds_bpermute_b32 v46,v62,v26\nds_bpermute_b32 v47,v63,v27\ns_waitcnt lgkmcnt(0)")
KERNEL(k_bperm_same_sel, 8, 8, "ds_bpermute_b32 v40,v56,v20\nds_bpermute_b32 v41,v56,v21\nds_bpermute_b32 v42,v56,v22\nds_bpermute_b32 v43,v56,v23\n"
                               "ds_bpermute_b32 v44,v56,v24\nds_bpermute_b32 v45,v56,v25\nds_bpermute_b32 v46,v56,v26\nds_bpermute_b32 v47,v56,v27\ns_waitcnt lgkmcnt(0)")
KERNEL(k_read_b32, 8, 8, "ds_read_b32 v40,v56\nds_read_b32 v41,v57\nds_read_b32 v42,v58\nds_read_b32 v43,v59\n"
                         "ds_read_b32 v44,v60\nds_read_b32 v45,v61\nds_read_b32 v46,v62\nds_read_b32 v47,v63\ns_waitcnt lgkmcnt(0)")
KERNEL(k_read_b128, 8, 4, "ds_read_b128 v[36:39],v56\nds_read_b128 v[40:43],v57\nds_read_b128 v[44:47],v58\nds_read_b128 v[48:51],v59\ns_waitcnt lgkmcnt(0)")
KERNEL(k_read_b128_b96, 8, 4, "ds_read_b128 v[36:39],v56\nds_read_b128 v[40:43],v57\nds_read_b96 v[44:46],v58\nds_read_b96 v[48:50],v59\ns_waitcnt lgkmcnt(0)")
KERNEL(k_read_b64, 8, 8, "ds_read_b64 v[36:37],v56\nds_read_b64 v[38:39],v57\nds_read_b64 v[40:41],v58\nds_read_b64 v[42:43],v59\n"
                         "ds_read_b64 v[44:45],v60\nds_read_b64 v[46:47],v61\nds_read_b64 v[48:49],v62\nds_read_b64 v[50:51],v63\ns_waitcnt lgkmcnt(0)")
KERNEL(k_read2st64, 8, 4, "ds_read2st64_b32 v[36:37],v56 offset0:0 offset1:1\nds_read2st64_b32 v[38:39],v56 offset0:2 offset1:3\nds_read2st64_b32 v[40:41],v56 offset0:4 offset1:5\nds_read2st64_b32 v[42:43],v56 offset0:6 offset1:7\ns_waitcnt lgkmcnt(0)")
KERNEL(k_write_b32, 8, 8, "ds_write_b32 v56,v20\nds_write_b32 v57,v21\nds_write_b32 v58,v22\nds_write_b32 v59,v23\n"
                          "ds_write_b32 v60,v24\nds_write_b32 v61,v25\nds_write_b32 v62,v26\nds_write_b32 v63,v27\ns_waitcnt lgkmcnt(0)")
KERNEL(k_add_u32, 8, 8, "ds_add_u32 v56,v39\nds_add_u32 v57,v39\nds_add_u32 v58,v39\nds_add_u32 v59,v39\n"
                        "ds_add_u32 v60,v39\nds_add_u32 v61,v39\nds_add_u32 v62,v39\nds_add_u32 v63,v39\ns_waitcnt lgkmcnt(0)")
KERNEL(k_min_u64, 8, 8, "ds_min_u64 v56,v[20:21]\nds_min_u64 v57,v[22:23]\nds_min_u64 v58,v[24:25]\nds_min_u64 v59,v[26:27]\n"
                        "ds_min_u64 v60,v[28:29]\nds_min_u64 v61,v[30:31]\nds_min_u64 v62,v[32:33]\nds_min_u64 v63,v[34:35]\ns_waitcnt lgkmcnt(0)")
// the BOX operation's LDS traffic in its real proportions: 1 item read, 6 pulls, node record, 4 appends, 1 counter add
KERNEL(k_boxmix_aos, 4, 16, "ds_read_b32 v40,v63\ns_waitcnt lgkmcnt(0)\nds_bpermute_b32 v41,v60,v21\nds_bpermute_b32 v42,v60,v22\nds_read_b128 v[36:39],v56\nds_read_b128 v[44:47],v57\n"
                            "ds_bpermute_b32 v43,v60,v23\nds_bpermute_b32 v48,v60,v24\nds_bpermute_b32 v49,v60,v25\nds_bpermute_b32 v50,v60,v26\nds_read_b96 v[28:30],v58\nds_read_b96 v[32:34],v59\n"
                            "s_waitcnt lgkmcnt(0)\nds_write_b32 v61,v20\nds_write_b32 v61,v21\nds_write_b32 v62,v22\nds_write_b32 v62,v23\nds_add_u32 v63,v39\ns_waitcnt lgkmcnt(0)")

struct Op {
  const char *name;
  void (*fn)(Args);
  int ninst;      // wave-instructions per loop iteration
  int pattern;    // LDS address pattern
  bool lds;
};

int main(int argc, char **argv) {
  int iters = 2000, warm = 1;
  const char *only = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-i") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-k") && i + 1 < argc) only = argv[++i];
    else if (!strcmp(argv[i], "-W") && i + 1 < argc) warm = atoi(argv[++i]);   // untimed launches ahead of the timed one
  }
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clockRate %d kHz; iters %d, %d warm-up launches\n", prop.name, cus, prop.clockRate, iters, warm);
  printf("# ns/inst/SIMD = kernel wall time (events) / (instructions per wave x waves per SIMD); an LDS instruction's share of its CU's LDS pipeline is a quarter of it\n");
  printf("# memtime/inst = mean s_memtime ticks per wave / (instructions per wave x waves per SIMD); wave-busy = mean s_memrealtime span of a wave's loop / kernel wall time\n");
  printf("# chip G inst/s = CUs x 4 SIMDs / (ns/inst/SIMD)\n");
#define OP(k, pat, lds) {#k, k, k##_ninst, pat, lds}
  std::vector<Op> ops = {
      OP(k_add, 0, false), OP(k_mul, 0, false), OP(k_fma, 0, false), OP(k_max3, 0, false), OP(k_min, 0, false),
      OP(k_add_dep, 0, false), OP(k_add_dep2, 0, false), OP(k_pk_mul, 0, false), OP(k_pk_add, 0, false),
      OP(k_cndmask_vcc, 0, false), OP(k_cndmask_sgpr, 0, false), OP(k_cndmask_vcc_e64, 0, false), OP(k_cmp_cnd_vcc, 0, false), OP(k_cmp_cnd_sgpr, 0, false), OP(k_sub, 0, false), OP(k_bfe, 0, false),
      OP(k_and, 0, false), OP(k_add_u32v, 0, false), OP(k_cmp_vcc, 0, false), OP(k_cmp_sgpr, 0, false),
      OP(k_cmp_salu_cnd, 0, false), OP(k_rcp, 0, false), OP(k_sqrt, 0, false), OP(k_div_scale, 0, false),
      OP(k_div_fmas, 0, false), OP(k_div_fixup, 0, false), OP(k_mbcnt, 0, false), OP(k_lshl_add, 0, false),
      OP(k_bfi, 0, false), OP(k_xor, 0, false), OP(k_or, 0, false), OP(k_lshlrev, 0, false), OP(k_ashrrev, 0, false), OP(k_max, 0, false),
      OP(k_med3, 0, false), OP(k_perm, 0, false), OP(k_and_or, 0, false), OP(k_lshl_or, 0, false), OP(k_add3, 0, false), OP(k_sub_u32, 0, false),
      OP(k_min_u32, 0, false), OP(k_fmac, 0, false), OP(k_pk_fma, 0, false), OP(k_pk_mov, 0, false),
      OP(k_mov, 0, false), OP(k_readfirstlane, 0, false), OP(k_salu, 0, false), OP(k_bcnt, 0, false), OP(k_snop, 0, false),
      OP(k_mix_v1s1, 0, false), OP(k_mix_v2s1, 0, false),
      {"k_bperm/identity", k_bperm, k_bperm_ninst, 9, true}, {"k_bperm/same-lane", k_bperm, k_bperm_ninst, 1, true},
      {"k_bperm/random", k_bperm, k_bperm_ninst, 2, true}, {"k_bperm/runs", k_bperm, k_bperm_ninst, 8, true},
      {"k_bperm_same_sel/runs", k_bperm_same_sel, k_bperm_same_sel_ninst, 8, true},
      {"k_read_b32/linear", k_read_b32, k_read_b32_ninst, 9, true}, {"k_read_b32/same", k_read_b32, k_read_b32_ninst, 1, true},
      {"k_read_b64/linear16", k_read_b64, k_read_b64_ninst, 0, true},
      {"k_read_b128/linear", k_read_b128, k_read_b128_ninst, 0, true}, {"k_read_b128/aos-random-record", k_read_b128, k_read_b128_ninst, 3, true},
      {"k_read_b128/planes-random-record", k_read_b128, k_read_b128_ninst, 4, true},
      {"k_read_b128/planes-sibling-pairs", k_read_b128, k_read_b128_ninst, 10, true},
      {"k_read_b128_b96/aos-random-record", k_read_b128_b96, k_read_b128_b96_ninst, 3, true},
      {"k_read_b32/raytable-planes", k_read_b32, k_read_b32_ninst, 11, true}, {"k_read2st64/raytable-planes", k_read2st64, k_read2st64_ninst, 11, true},
      {"k_read2st64/linear", k_read2st64, k_read2st64_ninst, 9, true},
      {"k_read_b64/linear8", k_read_b64, k_read_b64_ninst, 12, true}, {"k_read_b128/rayrecords", k_read_b128, k_read_b128_ninst, 13, true},
      {"k_write_b32/linear", k_write_b32, k_write_b32_ninst, 9, true}, {"k_write_b32/same", k_write_b32, k_write_b32_ninst, 1, true},
      {"k_write_b32/append+shared-dump", k_write_b32, k_write_b32_ninst, 5, true},
      {"k_write_b32/append+lane-dump", k_write_b32, k_write_b32_ninst, 6, true},
      {"k_add_u32/linear", k_add_u32, k_add_u32_ninst, 9, true}, {"k_add_u32/same", k_add_u32, k_add_u32_ninst, 1, true},
      {"k_add_u32/slot-runs", k_add_u32, k_add_u32_ninst, 7, true}, {"k_add_u32/random-slot", k_add_u32, k_add_u32_ninst, 2, true},
      {"k_min_u64/slot-runs", k_min_u64, k_min_u64_ninst, 7, true}, {"k_min_u64/same", k_min_u64, k_min_u64_ninst, 1, true},
      {"k_boxmix_aos/aos-random-record", k_boxmix_aos, k_boxmix_aos_ninst, 3, true},
      {"k_boxmix_aos/planes-random-record", k_boxmix_aos, k_boxmix_aos_ninst, 4, true},
  };
  unsigned long long *cyc;
  int *sink;
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 65536));
  CHECK(hipMalloc(&sink, 64));
  std::vector<unsigned long long> h(65536);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-40s %6s %14s %12s %14s %12s\n", "op", "w/SIMD", "ns/inst/SIMD", "chip Ginst/s", "memtime/inst", "wave-busy");
  for (const Op &op : ops) {
    if (only) {   // -k matches the part of the op name before its '/'
      const size_t n = strlen(only);
      if (strncmp(op.name, only, n) != 0 || (op.name[n] != '\0' && op.name[n] != '/')) continue;
    }
    CHECK(hipFuncSetAttribute((const void *)op.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int w : {1, 2, 4, 8}) {
      const int threads = w <= 4 ? 256 * w : 1024;
      const int blocks = (w <= 4 ? 1 : 2) * cus;
      const int it = op.lds ? iters / 4 + 1 : iters;
      Args a{cyc, sink, it, op.pattern};
      for (int k = 0; k < warm; ++k) hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(threads), 65536, 0, a);   // warm-up
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(threads), 65536, 0, a);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const int nw = blocks * (threads / 64);
      CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 65536, hipMemcpyDeviceToHost));
      double sum = 0, sum_real = 0;
      for (int i = 0; i < nw; ++i) { sum += (double)h[i]; sum_real += (double)h[32768 + i]; }
      const double mean = sum / nw, mean_real = sum_real / nw;   // s_memrealtime: 100 MHz
      const double ninst = (double)op.ninst * it;
      const double cpi = mean / (ninst * w);
      // wall-clock figure: the kernel is nothing but the loop (LDS fill + launch ~10 us of >= 500 us)
      const double ns = ms * 1e6 / (ninst * w);
      printf("%-40s %6d %14.3f %12.1f %14.3f %12.3f\n", op.name, w, ns, cus * 4 / ns, cpi, mean_real * 10.0 / (ms * 1e6));
      fflush(stdout);
    }
  }
  return 0;
}
