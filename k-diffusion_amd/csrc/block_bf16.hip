// Rows-stationary block kernels of the bf16 mode (round 5), gfx950: a workgroup owns 256 rows of ONE sample, normalises them once into MFMA B
// fragments held in registers, and streams 64-row half blocks of a packed weight past them.
//
//   kd_attn_block_bf16   the front of a global-attention block at 256 tokens per sample -- AdaRMSNorm -> qkv projection of one head ->
//                        cosine-sim scale + RoPE -> dense attention (the core of attn_bf16.hip) -- in one launch per layer; q, k, v never
//                        reach HBM.  On request the block's out projection + residual in the same launch (per-sample rendezvous).
//                        (image_transformer_v2.py:370-396)
//   kd_proj_block_bf16   AdaRMSNorm -> up projection + GEGLU (:487-491) and AdaRMSNorm -> qkv projection + cosine-sim scale + RoPE (for the
//                        levels whose attention core is its own launch) in the same form.
// Both reproduce the A-stationary projection kernel (gemm_bf16.hip) / the dense core bit for bit.
#include "attn_bf16_core.h"

namespace kd {
namespace b16 {

// ---- global-attention block in ONE launch: AdaRMSNorm -> qkv projection of one head -> cosine-sim + RoPE -> dense attention ----------
// (image_transformer_v2.py:370-396: norm, qkv_proj, scale_for_cosine_sim_qkv, apply_rotary_emb_, attention -- everything in front of
// out_proj.)  At the level-2 shape (8192 rows, K = 512) the two-launch form spent 29.5 us in the projection (a 4 us K loop per wave
// between a row prologue repeated per n-split and the q / k epilogues) and 13.5 us in a core that has 1.7 us of work, with a 25 MB
// qkv round trip through HBM between them.  Here a workgroup owns ONE (sample, head) problem with T = 256 tokens:
//   * its 8 waves take 32 rows each: the sample's rows come in by LDS-DMA once per workgroup, are normalised and scaled into
//     MFMA B-operand fragments held in registers (the A-stationary form of csrc/gemm_bf16.hip, same arithmetic, same order);
//   * the head's 64 rows of W_k, W_v, W_q stream past them (three passes over K, 8 KiB half blocks of the packed image, 4-slot ring of
//     two k-steps each, one barrier per 16 MFMAs per wave); k and v leave their epilogues as bf16 rows of the K / V images in LDS
//     (the dense core's swizzle), q -- the last pass -- stays in registers as the B fragments of S^T = K Q^T;
//   * then the dense core of attn_bf16.hip (attn_dense_bf16_kernel), unchanged: scores, softmax, O^T = V^T P^T, 16-byte stores of the attention output.
// Neither q, k nor v ever reaches HBM.  The arithmetic is that of the two-launch form operation for operation (the results are
// bit-identical: tests/test_ops_gpu.py::test_attn_block_bf16_matches_two_launches).  144 KiB of LDS, one workgroup per CU.
struct BArgs {
  const u16* x; const char* Wp; u16* out;
  const float* scale; int scale_stride; float eps;
  int batch, nh;                      // 256 tokens per sample
  const float* qk_scale; const float* pos; const float* freq;
  int warm;
  unsigned long long* clk;            // kd_prof_clock_buffer (TS instantiation only): workgroup 0's time line, s_memtime stamps
};
extern unsigned long long* g_clk;     // gemm_bf16.hip

__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  }
}

// one 64-dim row held as two C-layout blocks -> its 128-byte row of a [token][64] bf16 image in LDS (chunk c at c ^ asw(row))
__device__ __forceinline__ void row_to_image(char* img, int rowi, const f32x16& b0, const f32x16& b1, float mul, int lh) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    unsigned pk[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x16& v = e ? b1 : b0;
      pk[g][0] = pack_bf16(v[4 * g] * mul, v[4 * g + 1] * mul);
      pk[g][1] = pack_bf16(v[4 * g + 2] * mul, v[4 * g + 3] * mul);
    }
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {
      half_swap(pk[gp][0], pk[gp + 1][0]);
      half_swap(pk[gp][1], pk[gp + 1][1]);
      const int c = 4 * e + gp + lh;                       // dims 8 c .. 8 c + 7 of the row
      *reinterpret_cast<u32x4*>(img + rowi * 128 + ((c ^ asw(rowi)) << 4)) = u32x4{pk[gp][0], pk[gp][1], pk[gp + 1][0], pk[gp + 1][1]};
    }
  }
}

// TS: in-kernel time line of workgroup 0 (kd_prof_clock_buffer; a template flag so that the measured kernel's loops stay what they are):
// [0] entry, [4] rows normalised, [5] / [6] / [8] end of the k / v / q pass, [9] scores + softmax done, [2] exit, [1] / [3] s_memrealtime.
// The 32 rows of one wave (rows 32 wid .. of sample b) -> normalised, scaled MFMA B fragments `a[NC]` held in registers, and the rows' RMS
// factor `rs`.  gemm_astat_kernel's arithmetic (same products, same order of the sum of squares), another staging schedule: the rows come in by
// K-HALVES of 256 elements -- all 32 rows x 512 bytes per round into the wave's 16 KiB slot, every lane reads its own row's 16 chunks (at
// K = 512 the row-halves schedule of the projection kernel leaves half the lanes idle in each round) -- and the second half is in flight while
// the first is converted; scales two chunks at a time, one pair ahead (32 registers of scales: the block kernels carry 2 waves per SIMD).
// A MACRO, not a function: as a forceinline function taking / returning the fragments the same code compiled to 40 more registers and, at
// K = 512, 600 bytes of scratch per lane.  Uses the enclosing kernel's smem, wid, lane, l31, lh and the constants NC, K; declares a, rs.
// ROW0: first of the workgroup's 256 rows.
#define KD_ROWS_TO_FRAGMENTS(XPTR, ROW0, SVEC, EPS) \
  bf16x8 a[NC]; \
  float rs; \
  { \
    constexpr int NH = K / 256; \
    char* stage = smem + wid * WBLK; \
    char* scl = smem + 8 * WBLK + wid * (K * 4); \
    const char* ssrc = reinterpret_cast<const char*>((SVEC)) + lane * 16; \
_Pragma("unroll") \
    for (int i = 0; i < K * 4 / 1024; ++i) glds16(ssrc + i * 1024, scl + i * 1024); \
    auto request = [&](int h) { \
_Pragma("unroll") \
      for (int i = 0; i < 16; ++i) { \
        const int rr = (i * 64 + lane) >> 5, qs = (i * 64 + lane) & 31; \
        const size_t grow = (size_t)(ROW0) + wid * 32 + rr; \
        glds16(reinterpret_cast<const char*>((XPTR) + grow * K) + h * 512 + ((qs ^ (rr & 15)) << 4), stage + i * 1024); \
      } \
    }; \
    float ssq = 0.f; \
    const float* spl = reinterpret_cast<const float*>(scl) + 8 * lh; \
    const char* rowp = stage + l31 * 512; \
    request(0); \
_Pragma("unroll") \
    for (int h = 0; h < NH; ++h) { \
      u32x4 raw[16]; \
      KD_WAIT_VM(0); \
_Pragma("unroll") \
      for (int c = 0; c < 16; ++c) raw[c] = *reinterpret_cast<const u32x4*>(rowp + (((2 * c + lh) ^ (l31 & 15)) << 4)); \
      if (h + 1 < NH) { \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        request(h + 1); \
      } \
      f32x4 s0[2][2], s1[2][2]; \
      auto load_scales = [&](int c0, int g) { \
_Pragma("unroll") \
        for (int u = 0; u < 2; ++u) { \
          s0[g][u] = *reinterpret_cast<const f32x4*>(spl + 16 * (16 * h + c0 + u)); \
          s1[g][u] = *reinterpret_cast<const f32x4*>(spl + 16 * (16 * h + c0 + u) + 4); \
        } \
      }; \
      load_scales(0, 0); \
      __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
      for (int c0 = 0; c0 < 16; c0 += 2) { \
        const int g = (c0 >> 1) & 1; \
        if (c0 + 2 < 16) load_scales(c0 + 2, g ^ 1); \
_Pragma("unroll") \
        for (int u = 0; u < 2; ++u) { \
          float x[8]; \
_Pragma("unroll") \
          for (int e = 0; e < 4; ++e) { x[2 * e] = bf_lo(raw[c0 + u][e]); x[2 * e + 1] = bf_hi(raw[c0 + u][e]); } \
_Pragma("unroll") \
          for (int e = 0; e < 8; ++e) ssq = fmaf(x[e], x[e], ssq); \
          u32x4 o = {pack_bf16(x[0] * s0[g][u][0], x[1] * s0[g][u][1]), pack_bf16(x[2] * s0[g][u][2], x[3] * s0[g][u][3]), \
                     pack_bf16(x[4] * s1[g][u][0], x[5] * s1[g][u][1]), pack_bf16(x[6] * s1[g][u][2], x[7] * s1[g][u][3])}; \
          asm volatile("" : "+v"(o)); \
          a[16 * h + c0 + u] = __builtin_bit_cast(bf16x8, o); \
        } \
        __builtin_amdgcn_sched_barrier(0); \
      } \
    } \
    ssq += __shfl_xor(ssq, 32, 64); \
    rs = rsqrtf(ssq / (float)K + (EPS)); \
  }

template <int NC /* K / 16 */, bool TS = false>
__global__ __launch_bounds__(512, 1) void attn_block_bf16_kernel(const BArgs p) {
  constexpr int K = NC * 16, NK = NC / 4, SPP = NK / 2, NSTAGE = 3 * SPP, NSLOT = 4, PDIST = 3, T = 256, NT = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kimg = smem + NSLOT * WBLK;
  char* Vimg = Kimg + T * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<24 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  const bool probe = TS && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }
  // extended form (x3_common.h: wg_stamp_begin): every workgroup's entry / exit in 100 MHz ticks at clk[32 + 3 b], + 1
  unsigned long long* wg_slot = (TS && tid == 0 && p.clk[15] == 0x4b44ull) ? p.clk + 32 + 3 * blockIdx.x : nullptr;
  if (TS && wg_slot) wg_slot[0] = __builtin_amdgcn_s_memrealtime();
  // workgroup -> (sample, head): ids go to the 8 XCDs round-robin, so the heads of one sample get ids 8 apart -- one XCD's L2 fetches the
  // sample's rows from HBM once for all of them
  int b, head;
  if ((p.batch & 7) == 0) {
    const int j = blockIdx.x >> 3;
    b = (j / p.nh) * 8 + (blockIdx.x & 7);
    head = j % p.nh;
  } else {
    b = blockIdx.x / p.nh;
    head = blockIdx.x % p.nh;
  }
  const int tok = wid * 32 + l31;                         // this lane's token of the sample: its row, later its query
  const size_t row = (size_t)b * T + tok;

  // ---- the wave's 32 rows -> normalised, scaled B fragments ---------------------------------------------------------------------------
  KD_ROWS_TO_FRAGMENTS(p.x, (size_t)b * T, p.scale + (size_t)b * p.scale_stride, p.eps)
  float py = p.pos[2 * tok], px = p.pos[2 * tok + 1];
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(py), "+v"(px) : : "memory");
  code_warm_end(warm);
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();

  // ---- the head's W_k, W_v, W_q rows past the fragments -------------------------------------------------------------------------------
  // stage s = pass s / SPP (k, v, q), k-steps 2 kk and 2 kk + 1 of that pass: two 8 KiB half blocks (rows 64 (head & 1) .. + 63 of the
  // block (n-tile of the head's rows, k-step)); wave w brings piece w of each
  const char* wbase = p.Wp + (head & 1) * 8192 + wid * 1024 + lane * 16;
  auto issue = [&](int s) {
    const int pass = s / SPP, kk = s % SPP;
    const int which = pass == 0 ? 1 : (pass == 1 ? 2 : 0);
    const int nt = (which * K + head * 64) >> 7;
    char* dst = smem + (s % NSLOT) * WBLK + wid * 1024;
    glds16(wbase + ((size_t)nt * NK + 2 * kk) * WBLK, dst);
    glds16(wbase + ((size_t)nt * NK + 2 * kk + 1) * WBLK, dst + 8192);
  };
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);

  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  // the head's constants through the scalar cache (see gemm_astat_kernel)
  typedef float f32x8s __attribute__((ext_vector_type(8)));
  f32x8s fq;
  float qsc;
  asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(fq), "=s"(qsc) : "s"(p.freq + head * 8), "s"(p.qk_scale + head) : "memory");
  float fr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) fr[u] = pick_half(fq[u], fq[4 + u], 0u - (unsigned)lh);
  const float sqs = sqrtf(qsc);

  bf16x8 qf[4];
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s) {
    const int pass = s / SPP, kk = s % SPP;
    if (kk == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    wait_vm_n(2 * min(PDIST - 1, NSTAGE - 1 - s));
    KD_BARRIER();                        // every wave's pieces of stage s are in; everyone is done reading slot (s - 1) % NSLOT
    if (s + PDIST < NSTAGE) issue(s + PDIST);
    const char* st = smem + (s % NSLOT) * WBLK;
    bf16x8 wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 128 + off4[0]);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {     // 8 chunks of 16 k: half block h = c8 / 4, chunk cc = c8 % 4
      if (c8 + 1 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[(c8 + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(st + ((c8 + 1) >> 2) * 8192 + j * 32 * 128 + off4[(c8 + 1) & 3]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c8 & 1][j], a[4 * (2 * kk + (c8 >> 2)) + (c8 & 3)], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk == SPP - 1) {
      if (pass == 0) {                   // k: cosine-sim scale + RoPE, then its row of the K image
        qk_prep_blocks(acc[0], acc[1], rs, sqs, p.eps, py, px, fr);
        row_to_image(Kimg, tok, acc[0], acc[1], 1.0f, lh);
      } else if (pass == 1) {            // v
        row_to_image(Vimg, tok, acc[0], acc[1], rs, lh);
      } else {                           // q: stays in registers as the B fragments of the score products (dims 16 st + 8 lh .. + 7)
        qk_prep_blocks(acc[0], acc[1], rs, sqs, p.eps, py, px, fr);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            pk[g][0] = pack_bf16(acc[e][4 * g], acc[e][4 * g + 1]);
            pk[g][1] = pack_bf16(acc[e][4 * g + 2], acc[e][4 * g + 3]);
          }
#pragma unroll
          for (int gp = 0; gp < 4; gp += 2) {
            half_swap(pk[gp][0], pk[gp + 1][0]);
            half_swap(pk[gp][1], pk[gp + 1][1]);
            qf[2 * e + gp / 2] = __builtin_bit_cast(bf16x8, u32x4{pk[gp][0], pk[gp][1], pk[gp + 1][0], pk[gp + 1][1]});
          }
        }
      }
      if (probe) p.clk[pass == 0 ? 5 : (pass == 1 ? 6 : 8)] = __builtin_amdgcn_s_memtime();
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  KD_BARRIER();                                        // all 256 rows of the K and V images are written

  // ---- S^T = K Q^T, softmax, O^T = V^T P^T: the dense core ------------------------------------------------------------------------------
  f32x16 S[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
  const int ka = l31 * 128 + ((lh ^ asw(l31)) << 4);
#pragma unroll
  for (int st = 0; st < 4; ++st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + (ka ^ (32 * st)) + 4096 * t);
      S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], S[t], 0, 0, 0);
    }
  }
  const float m = score_max<NT>(S);
  float l = score_exp<NT>(S, m);
  l += __shfl_xor(l, 32, 64);
  if (probe) p.clk[9] = __builtin_amdgcn_s_memtime();
  f32x16 O[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[e][i] = 0.f;
  const int va = vt_addr(4 * lh + vt_lane_row(lane), lane);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) pv_step(O, Vimg + (32 * t + 16 * u) * 128, va, p_frag(S[t], u));
  store_o(p.out + row * (size_t)(p.nh * DH) + head * DH, O, 1.0f / l, lh, true);
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); }
  if (TS && wg_slot) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wg_slot[1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- AdaRMSNorm -> wide projection (GEGLU up projection, qkv) in the attention block's form ------------------------------------------------
// (image_transformer_v2.py:487-491: norm, up_proj / linear_geglu; :370-380 / :415-425: norm, qkv_proj, cosine-sim scale, RoPE -- for the levels
// whose attention core is a separate launch.)  The A-stationary projection kernel (gemm_bf16.hip: gemm_astat_kernel) runs these shapes --
// 8 192 rows x K = 512 x 3 072 W rows, 32 768 rows x K = 256 x 1 536 / 768 W rows -- as 128-row panels x n-splits: every split repeats the
// panel's row prologue, and two workgroups per CU re-stream up to 1 MiB of rows + weights through a 37 - 50 bytes / clock L2 -> LDS path.
// Here a workgroup owns (256-row group, slice of six 64-row half blocks of the packed image): its 8 waves normalise the group's rows ONCE into
// register fragments (KD_ROWS_TO_FRAGMENTS) and then run six passes over K through the attention block's 4-slot ring; a pass ends with the
// epilogue of the lane's own row (GEGLU of 32 value / 32 gate columns; cosine-sim scale + RoPE of a q / k head vector; the row factor for v) and
// its 16-byte stores.  Same products in the same order as gemm_astat_kernel<NC, EPI>: bit-identical.
struct UArgs {
  const u16* x; const char* Wp; u16* out;
  const float* scale; int scale_stride; float eps;
  int groups, groups_per_sample, slices, n_out;      // 256-row groups; groups of one sample; slices of 6 half blocks; output row width
  int n_heads; const float* qk_scale; const float* pos; const float* freq;      // EPI_QKV
  int warm;
};

__device__ __forceinline__ void wait_vm_any(int n) {
  switch (n) {
#define KD_C(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
    KD_C(0) KD_C(1) KD_C(2) KD_C(3) KD_C(4) KD_C(5) KD_C(6) KD_C(7) KD_C(8) KD_C(9) KD_C(10) KD_C(11) KD_C(12) KD_C(13) KD_C(14) KD_C(15) KD_C(16)
#undef KD_C
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

template <int NC /* K / 16 */, int EPI>
__global__ __launch_bounds__(512, 1) void proj_block_bf16_kernel(const UArgs p) {
  constexpr int K = NC * 16, NK = NC / 4, SPP = NK / 2, NPASS = 6, NSTAGE = NPASS * SPP, NSLOT = 4, PDIST = 3;
  constexpr int NST = EPI == KD_EPI_GEGLU ? 2 : 4;      // 16-byte stores per lane at the end of a pass
  constexpr int COLS = EPI == KD_EPI_GEGLU ? 32 : 64;   // output columns of a half block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<24 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  int grp, slice;                                        // the slices of one row group on one XCD (ids 8 apart), as the attention block's heads
  if ((p.groups & 7) == 0) {
    const int j = blockIdx.x >> 3;
    grp = (j / p.slices) * 8 + (blockIdx.x & 7);
    slice = j % p.slices;
  } else {
    grp = blockIdx.x / p.slices;
    slice = blockIdx.x % p.slices;
  }
  const int b = grp / p.groups_per_sample;
  const size_t row0 = (size_t)grp * 256;
  const size_t row = row0 + wid * 32 + l31;
  KD_ROWS_TO_FRAGMENTS(p.x, row0, p.scale + (size_t)b * p.scale_stride, p.eps)
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    const int tok = (int)(row - (size_t)b * p.groups_per_sample * 256);
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(py), "+v"(px) : : "memory");
  }
  code_warm_end(warm);
  KD_BARRIER();                                        // every wave has taken its rows out of the slot it borrowed

  // stage s = pass s / SPP (half block 6 slice + pass), k-steps 2 kk and 2 kk + 1; wave w brings piece w of each half block
  const char* wbase = p.Wp + wid * 1024 + lane * 16;
  auto issue = [&](int s) {
    const int hb = NPASS * slice + s / SPP, kk = s % SPP;
    const char* src = wbase + ((size_t)(hb >> 1) * NK + 2 * kk) * WBLK + (hb & 1) * 8192;
    char* dst = smem + (s % NSLOT) * WBLK + wid * 1024;
    glds16(src, dst);
    glds16(src + WBLK, dst + 8192);
  };
#pragma unroll
  for (int s = 0; s < PDIST; ++s) issue(s);
  int off4[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) off4[cc] = swz128(l31, 2 * cc + lh);
  u16* crow = p.out + row * (size_t)p.n_out + (size_t)NPASS * slice * COLS;
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s) {
    const int pass = s / SPP, kk = s % SPP;
    if (kk == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    {
      // behind stage s in the queue (loads and stores retire in issue order): the stages requested after it and the stores of every pass
      // that ended since its request (iterations s - PDIST .. s - 1)
      int allow = 2 * min(PDIST - 1, NSTAGE - 1 - s);
#pragma unroll
      for (int e = s - PDIST; e <= s - 1; ++e)
        if (e >= 0 && e % SPP == SPP - 1) allow += NST;
      wait_vm_any(allow);
    }
    KD_BARRIER();
    if (s + PDIST < NSTAGE) issue(s + PDIST);
    const char* st = smem + (s % NSLOT) * WBLK;
    bf16x8 wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(st + j * 32 * 128 + off4[0]);
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      if (c8 + 1 < 8) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[(c8 + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(st + ((c8 + 1) >> 2) * 8192 + j * 32 * 128 + off4[(c8 + 1) & 3]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c8 & 1][j], a[4 * (2 * kk + (c8 >> 2)) + (c8 & 3)], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk == SPP - 1) {
      if (EPI == KD_EPI_GEGLU) {                        // value block acc[0], gate block acc[1] (gemm_astat_kernel's epilogue)
        float v[16];
        const float rsh = 0.5f * rs;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 o = geglu_pair(f32x2{acc[0][r], acc[0][r + 1]} * rsh, f32x2{acc[1][r], acc[1][r + 1]} * rs);
          v[r] = o.x;
          v[r + 1] = o.y;
        }
        store_block_bf16(crow + 32 * pass, v, lh, true);
      } else {                                          // one 64-column vector of q, k or v: dims 0..31 in acc[0], 32..63 in acc[1]
        const int vec = NPASS * slice + pass;
        const int which = vec / p.n_heads, head = vec - which * p.n_heads;
        if (which < 2) {
          typedef float f32x8s __attribute__((ext_vector_type(8)));
          f32x8s fq;
          float qsc;
          asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                       : "=s"(fq), "=s"(qsc) : "s"(p.freq + head * 8), "s"(p.qk_scale + head) : "memory");
          float fr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) fr[u] = pick_half(fq[u], fq[4 + u], 0u - (unsigned)lh);
          qk_prep_blocks(acc[0], acc[1], rs, sqrtf(qsc), p.eps, py, px, fr);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[0][r] *= rs; acc[1][r] *= rs; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[jj][r];
          store_block_bf16(crow + 64 * pass + 32 * jj, v, lh, true);
        }
      }
    }
  }
}

}  // namespace b16
}  // namespace kd

using namespace kd;
using namespace kd::b16;

// The global-attention block in front of its out projection as ONE launch (attn_block_bf16_kernel above).  `d` is the descriptor of the
// block's qkv projection exactly as kd_gemm_bf16 takes it (A = the residual stream, Wp = the packed qkv weight, scale / scale_stride /
// rows_per_sample = the AdaRMSNorm scale table, qk_scale / rope_pos / rope_freq / n_heads) -- except that C receives the ATTENTION OUTPUT
// [M, n_heads * 64] bf16 instead of qkv.  Shapes: 256 tokens per sample, K = n_heads * 64 in {256, 512}, N = 3 K.
extern "C" int kd_attn_block_bf16_supported(int tokens_per_sample, int width, int n_heads) {
  return tokens_per_sample == 256 && (width == 256 || width == 512) && n_heads * 64 == width && option("attn_block_bf16", 1) ? 1 : 0;
}

extern "C" int kd_attn_block_bf16(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_attn_block_bf16: null descriptor");
  const KdGemm& d = *dp;
  if (!d.A || !d.Wp || !d.C || !d.scale || !d.qk_scale || !d.rope_pos || !d.rope_freq) return fail(KD_EINVAL, "kd_attn_block_bf16: null operand");
  if (d.epi != KD_EPI_QKV || !d.norm || d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16)
    return fail(KD_EINVAL, "kd_attn_block_bf16: the descriptor must be a bf16 norm -> qkv projection");
  if (!kd_attn_block_bf16_supported(d.rows_per_sample, d.K, d.n_heads) || d.N != 3 * d.K || d.M <= 0 || d.M % 256)
    return fail(KD_EINVAL, "kd_attn_block_bf16: shape M=%d N=%d K=%d, %d tokens per sample, %d heads is not taken (256 tokens per sample, "
                "K = 64 heads in {256, 512})", d.M, d.N, d.K, d.rows_per_sample, d.n_heads);
  BArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp), reinterpret_cast<u16*>(d.C), d.scale, d.scale_stride, d.eps,
          d.M / 256, d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT), g_clk};
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS = 8 * WBLK + 8 * 512 * 4;          // prologue: 8 wave-private staging slots + 8 scale vectors; later ring + K / V images
  const double flops = 2.0 * d.M * 3.0 * d.K * d.K + 4.0 * (double)a.batch * a.nh * 256.0 * 256.0 * DH;
  const double bytes = 2.0 * ((double)d.M * d.K * 2.0 + 3.0 * d.K * d.K);
  char nm[96] = "attn_block_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "attn_block_bf16 M=%d K=%d nh=%d", d.M, d.K, d.n_heads);
  LaunchScope prof(nm, flops, bytes, s);
#define KD_BLK(NCV, TSV) { static LdsAttr set; set.ensure(reinterpret_cast<const void*>(attn_block_bf16_kernel<NCV, TSV>), LDS); \
    hipLaunchKernelGGL((attn_block_bf16_kernel<NCV, TSV>), dim3((unsigned)(a.batch * a.nh)), dim3(512), LDS, s, a); }
#define KD_BLK2(NCV) { if (a.clk) KD_BLK(NCV, true) else KD_BLK(NCV, false) }
  if (d.K == 512) KD_BLK2(32) else KD_BLK2(16)
#undef KD_BLK2
#undef KD_BLK
  return check_launch("kd_attn_block_bf16");
}

// AdaRMSNorm -> wide projection in the attention block's form (proj_block_bf16_kernel above).  `d` is the projection's descriptor exactly as
// kd_gemm_bf16 takes it -- epi = KD_EPI_GEGLU (N = d_ff) or KD_EPI_QKV (N = 3 K; the qkv tensor is written, for the levels whose attention core
// is its own launch), norm = 1 -- and the results are bit-identical to that call.  Rows per sample a multiple of 256, K in {256, 512}, the W rows
// a multiple of 6 half blocks (d_ff % 192 == 0; 3 K % 384 == 0 holds for both widths).
extern "C" int kd_proj_block_bf16_supported(int tokens_per_sample, int width, int n, int epi) {
  if (tokens_per_sample <= 0 || tokens_per_sample % 256 || (width != 256 && width != 512) || !option("proj_block_bf16", 1)) return 0;
  if (epi == KD_EPI_GEGLU) return n > 0 && n % 192 == 0;
  if (epi == KD_EPI_QKV) return n == 3 * width;
  return 0;
}

extern "C" int kd_proj_block_bf16(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_proj_block_bf16: null descriptor");
  const KdGemm& d = *dp;
  if (!d.A || !d.Wp || !d.C || !d.scale) return fail(KD_EINVAL, "kd_proj_block_bf16: null operand");
  if ((d.epi != KD_EPI_GEGLU && d.epi != KD_EPI_QKV) || !d.norm || d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16)
    return fail(KD_EINVAL, "kd_proj_block_bf16: the descriptor must be a bf16 norm -> GEGLU or norm -> qkv projection");
  if (d.epi == KD_EPI_QKV && (!d.qk_scale || !d.rope_pos || !d.rope_freq || d.n_heads * 64 != d.K))
    return fail(KD_EINVAL, "kd_proj_block_bf16: the qkv projection needs qk_scale, rope_pos, rope_freq and n_heads * 64 == K");
  if (!kd_proj_block_bf16_supported(d.rows_per_sample, d.K, d.N, d.epi) || d.M <= 0 || d.M % d.rows_per_sample)
    return fail(KD_EINVAL, "kd_proj_block_bf16: shape M=%d N=%d K=%d, %d tokens per sample is not taken (tokens per sample a multiple of 256, K in "
                "{256, 512}, d_ff a multiple of 192)", d.M, d.N, d.K, d.rows_per_sample);
  const int w_rows = d.epi == KD_EPI_GEGLU ? 2 * d.N : d.N;
  UArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp), reinterpret_cast<u16*>(d.C), d.scale, d.scale_stride, d.eps,
          d.M / 256, d.rows_per_sample / 256, w_rows / 384, d.N, d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT)};
  hipStream_t s = (hipStream_t)stream;
  constexpr int LDS = 8 * WBLK + 8 * 512 * 4;
  const double flops = 2.0 * d.M * (double)w_rows * d.K;
  const double bytes = 2.0 * ((double)d.M * d.K + (double)w_rows * d.K + (double)d.M * d.N);
  char nm[96] = "proj_block_bf16";
  if (prof_on()) snprintf(nm, sizeof(nm), "proj_block_bf16<e%d> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
  LaunchScope prof(nm, flops, bytes, s);
#define KD_PB(NCV, EP) { static LdsAttr set; set.ensure(reinterpret_cast<const void*>(proj_block_bf16_kernel<NCV, EP>), LDS); \
    hipLaunchKernelGGL((proj_block_bf16_kernel<NCV, EP>), dim3((unsigned)(a.groups * a.slices)), dim3(512), LDS, s, a); }
  if (d.K == 512) { if (d.epi == KD_EPI_GEGLU) KD_PB(32, KD_EPI_GEGLU) else KD_PB(32, KD_EPI_QKV) }
  else { if (d.epi == KD_EPI_GEGLU) KD_PB(16, KD_EPI_GEGLU) else KD_PB(16, KD_EPI_QKV) }
#undef KD_PB
  return check_launch("kd_proj_block_bf16");
}

KD_TEXT_PAD(block_bf16)      // last function of this code object: kd_common.h, code warm-up
