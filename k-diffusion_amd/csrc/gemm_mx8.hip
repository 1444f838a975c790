// fp8 arithmetic mode (KD_PREC_FP8; BASELINE configs[4] "fp8 MFMA weights"): the AdaRMSNorm -> wide projections of the K = 256 / 512 levels
// (qkv + cosine-sim + RoPE, up projection + GEGLU, plain store) on gfx950's block-scaled fp8 matrix instruction,
// v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate, 5 PFLOP/s dense).  There is no reference counterpart
// (convert_for_inference.py:23 stops at fp16 / bf16); the arithmetic is defined HERE and restated in oracle/hdit.py (mx8_*):
//
//   weights      W[n, :] = 2^e_n * q_n,  q_n OCP e4m3 (RNE), e_n = ceil(log2(max|W[n, :]| / 448)): one power-of-two scale per output
//                channel -- exactly checkpoint.quantize_fp8, so an fp8 checkpoint's weights enter the instruction bit for bit;
//   activations  u = x * s (x the bf16 residual row, s the sample's fp32 AdaRMSNorm scale vector) is quantised per (row, 32-k block):
//                u[b] = 2^E_b * q_b, E_b = ceil(log2(max|u[b]| / 448)) from the fp32 quotient's bits (no saturation: |q| <= 448), q OCP e4m3
//                (RNE) -- OCP microscaling, which the instruction takes natively: one E8M0 scale byte per (row, 32-k block);
//   products     exact in the instruction, fp32 accumulation; the RMS row factor rsqrt(mean(x^2) + eps) (fp32 statistics of the UNQUANTISED
//                row) multiplies the accumulators in the epilogue, which is gemm_bf16.hip's (cosine-sim norm, RoPE, GELU in fp32).
//
// Kernel form: gemm_bf16.hip's A-stationary projection (gemm_astat_kernel) with the matrix instruction and the operand formats changed:
// a wave keeps its 32 rows as fp8 B fragments in registers (K / 8 registers instead of K / 4: both widths leave room for two waves per SIMD),
// the packed fp8 weight streams through the same 4-slot ring of 16 KiB blocks -- a block now covers 128 k instead of 64 -- by LDS-DMA, and a
// k-step of 64 is ONE instruction per 32-feature group (8 per block, 64 cycles each) fed by two ds_read_b128.
//
// Packed image (kd_pack_weight_mx8): blocks [n-tile][k / 128] of 16 KiB, inside a block the 16-byte piece (feature group j of 32, k-step ks
// of 64, 32-k block h of the step = half h of a lane's 32 bytes, lane half lh = which 16 k of that block, feature l31) at
// ((((j * 2 + ks) * 2 + h) * 2 + lh) * 32 + l31) * 16: every ds_read_b128 of a W fragment half covers one contiguous KiB.  Behind the blocks: one E8M0 byte per tile row, [n-tile][l31][j].
// GEGLU tiles interleave 32 value rows with their 32 gate rows like the bf16 image (bf16_common.h: w_row_of_tile).
#include "bf16_common.h"

namespace kd {
namespace b16 {
extern unsigned long long* g_clk;     // gemm_bf16.hip
}
namespace mx8 {
using namespace b16;

using i32x8 = __attribute__((ext_vector_type(8))) int;

struct MArgs {
  const u16* A; const char* Wp; const unsigned* Ws; u16* C;
  unsigned char* C8; unsigned char* Cs;        // C8OUT (GEGLU): the result as e4m3 [M, N] + one E8M0 byte per (row, 32 features) [M, N / 32]
  const float* scale; int scale_stride, rows_per_sample; float eps;
  int M, N, n_tiles, n_splits;
  int n_heads; const float* qk_scale; const float* pos; const float* freq;
  int warm;
  unsigned long long* clk;  // kd_prof_clock_buffer: workgroup 0's time line (s_memtime): [0] entry, [4] rows quantised, [5] / [6] first tile's K loop / epilogue done, [2] exit
};

// E8M0 exponent byte of the power-of-two block scale 2^(E - 127) >= amax / 448 (smallest such), clamped to [1, 253]
// (the definition -- oracle/hdit.py: mx8_scale, the C++ harness -- reads it off the bits of the correctly rounded fp32 quotient amax / 448:
// exponent, + 1 unless the mantissa is zero.  448 = 1.75 * 2^8, so that is: exponent of amax - 8, + 1 iff its mantissa exceeds 1.75's -- the
// same byte for every float (checked over all 2^23 mantissas), without the ten instructions of an IEEE division per block.)
__device__ __host__ __forceinline__ unsigned mx_scale_byte(float amax) {
  const int b = (int)((__builtin_bit_cast(unsigned, amax) + 0x1FFFFFu) >> 23) - 8;
  return (unsigned)(b < 1 ? 1 : (b > 253 ? 253 : b));
}
__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }

__device__ __forceinline__ void glds16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ void wait_vm_dyn8(int n) {
  switch (n) {
#define KD_C(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
    KD_C(0) KD_C(1) KD_C(2) KD_C(3) KD_C(4) KD_C(5) KD_C(6) KD_C(7) KD_C(8) KD_C(9) KD_C(10) KD_C(11) KD_C(12) KD_C(13) KD_C(14) KD_C(15)
    KD_C(16) KD_C(17) KD_C(18) KD_C(19) KD_C(20) KD_C(21) KD_C(22) KD_C(23) KD_C(24) KD_C(25) KD_C(26) KD_C(27) KD_C(28) KD_C(29) KD_C(30) KD_C(31)
#undef KD_C
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// NK64 = K / 64 (4 or 8).  128-row panels (4 waves), two workgroups per CU, n-splits of a panel on one XCD: gemm_astat_kernel's schedule.
// C8OUT (EPI_GEGLU): the hidden activation leaves as the NEXT fp8 product's A operand -- e4m3 rows + one power-of-two scale per (row, 32
// features), the same rule as the activations above -- instead of bf16: kd_gemm_mx8's tiled form (down projection) takes it by LDS-DMA.
template <int NK64, int EPI, bool C8OUT = false>
__global__ __launch_bounds__(256, 2) void gemm_mx8_astat_kernel(const MArgs p) {
  static_assert(!C8OUT || EPI == KD_EPI_GEGLU, "e4m3 output: the GEGLU epilogue");
  constexpr int K = NK64 * 64, NKB = K / 128, NSTG = 4, PDIST = NSTG - 1, NWV = 4, PB = 4;
  constexpr bool GEGLU = EPI == KD_EPI_GEGLU;
  constexpr int NCOL = GEGLU ? 64 : 128;
  constexpr int NST = GEGLU ? 4 : 8;                 // 16-byte stores per lane per n-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<16 * 1024>((int)blockIdx.x < p.warm && tid < 64);
  int panel, split;
  const int n_splits = p.n_splits, n_panels = gridDim.x / n_splits;
  if ((n_panels & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    panel = (j / n_splits) * 8 + xcd;
    split = j % n_splits;
  } else {
    panel = blockIdx.x % n_panels;
    split = blockIdx.x / n_panels;
  }
  const int nt_begin = (int)((long)p.n_tiles * split / n_splits), nt_end = (int)((long)p.n_tiles * (split + 1) / n_splits);
  const int n_tiles = nt_end - nt_begin, total = n_tiles * NKB;
  const int m0 = panel * 128;
  const bool probe = p.clk && blockIdx.x == 0 && tid == 0;
  if (probe) { p.clk[0] = __builtin_amdgcn_s_memtime(); p.clk[1] = __builtin_amdgcn_s_memrealtime(); }

  const char* wp = p.Wp + (size_t)nt_begin * NKB * WBLK + wid * (PB * 1024) + lane * 16;
  auto issue = [&](int s) {
    const char* src = wp + (size_t)s * WBLK;
    char* dst = smem + (s % NSTG) * WBLK + wid * (PB * 1024);
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
  };

  const int row = m0 + wid * 32 + l31;
  const bool ok = row < p.M;
  const int rowc = ok ? row : p.M - 1;
  // The instruction's operand layout, found by experiment (benchmarks/probe/mx8_probe.cpp, profiles/r06_mx8_probe.log): lane (row, lh) holds,
  // of a 64-wide k-step, bytes 0-15 = k 16 lh .. 16 lh + 15 of the step's FIRST 32-k block and bytes 16-31 = the same k of its SECOND block;
  // block b's E8M0 byte is read from the scale register of lane (row, lh = b); op_sel picks the byte of that register.
  i32x8 a[NK64];                                       // this lane's 2 x 16 e4m3 values of every 64-wide k-step
  int asc[NK64];                                       // ... and the E8M0 byte of block lh of the step, in all four byte lanes (any op_sel reads it)
  float rs;
  {
    // rows: HBM -> the ring slot this wave borrows (whole rows by LDS-DMA) -> this lane's 64-byte pieces; the sample's scale vector -> LDS
    constexpr int RPR = WBLK / (2 * K);                // rows per staging round: 16 at K = 512, 32 at K = 256
    constexpr int NR = 32 / RPR, CPR = K / 8;
    static_assert(RPR >= 16 && NR * RPR == 32, "staging geometry");
    u32x4 raw[NK64][4];
    char* stage = smem + wid * WBLK;
    char* scl = smem + NSTG * WBLK + wid * (K * 4);
    const int r_first = min(m0 + wid * 32, p.M - 1), r_last = min(m0 + wid * 32 + 31, p.M - 1);
    const bool uni = r_first / p.rows_per_sample == r_last / p.rows_per_sample;
    const int b = rowc / p.rows_per_sample;
    const float* sp = p.scale + (size_t)b * p.scale_stride;
    if (uni) {
      const char* ssrc = reinterpret_cast<const char*>(p.scale + (size_t)(r_first / p.rows_per_sample) * p.scale_stride) + lane * 16;
#pragma unroll
      for (int i = 0; i < K * 4 / 1024; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + i * 1024),
                                         (__attribute__((address_space(3))) void*)(scl + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i * 64 + lane) / CPR, qs = (i * 64 + lane) % CPR;
        const int grow = min(m0 + wid * 32 + r * RPR + rr, p.M - 1);
        const char* src = reinterpret_cast<const char*>(p.A + (size_t)grow * K) + ((qs ^ (rr & 15)) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // wave-private slot: no barrier
      if (NR == 1 || (l31 / RPR) == r) {
        const int rr = l31 % RPR;
        const char* rowp = stage + rr * (2 * K);
#pragma unroll
        for (int ks = 0; ks < NK64; ++ks)
#pragma unroll
          for (int u = 0; u < 4; ++u)       // u = 2 * block + piece: 8 k of block (u >> 1), k = 16 lh + 8 (u & 1) .. inside it
            raw[ks][u] = *reinterpret_cast<const u32x4*>(rowp + (((8 * ks + 4 * (u >> 1) + 2 * lh + (u & 1)) ^ (rr & 15)) << 4));
      }
      if (NR > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // packed fp32 arithmetic throughout (two elements per v_pk_mul / v_pk_fma; the block maxima by v_max3 with |.| modifiers): this prologue is
    // vector-issue bound and every n-split of a panel repeats it
    f32x2 ssq2 = {0.f, 0.f};
    const float* spl = reinterpret_cast<const float*>(scl);
#pragma unroll
    for (int ks = 0; ks < NK64; ++ks) {
      f32x2 y[16];
      float amax[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k0 = 64 * ks + 32 * (u >> 1) + 16 * lh + 8 * (u & 1);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>((uni ? spl : sp) + k0), s1 = *reinterpret_cast<const f32x4*>((uni ? spl : sp) + k0 + 4);
        const f32x2 sc[4] = {{s0[0], s0[1]}, {s0[2], s0[3]}, {s1[0], s1[1]}, {s1[2], s1[3]}};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 x = {bf_lo(raw[ks][u][e]), bf_hi(raw[ks][u][e])};
          ssq2 = __builtin_elementwise_fma(x, x, ssq2);
          const f32x2 v = x * sc[e];
          y[4 * u + e] = v;
          amax[u >> 1] = max3f(amax[u >> 1], fabsf(v.x), fabsf(v.y));
        }
      }
      // a 32-k block lives in BOTH lanes of a row (16 values each): its maximum is the larger of the two halves'
      unsigned sb[2];
      f32x2 inv[2];
#pragma unroll
      for (int bk = 0; bk < 2; ++bk) {
        amax[bk] = fmaxf(amax[bk], __shfl_xor(amax[bk], 32, 64));
        sb[bk] = mx_scale_byte(amax[bk]);
        const float iv = mx_inv_scale(sb[bk]);
        inv[bk] = f32x2{iv, iv};
      }
      i32x8 f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const f32x2 q0 = y[2 * w] * inv[w >> 2], q1 = y[2 * w + 1] * inv[w >> 2];
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(q0.x, q0.y, pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(q1.x, q1.y, pk, true);
        f[w] = pk;
      }
      asm volatile("" : "+v"(f));                      // materialise the fragment here (see gemm_astat_kernel)
      a[ks] = f;
      asc[ks] = (int)((lh ? sb[1] : sb[0]) * 0x01010101u);      // block b's byte is read from the lane half b of the row
      __builtin_amdgcn_sched_barrier(0);
    }
    float ssq = ssq2.x + ssq2.y;
    ssq += __shfl_xor(ssq, 32, 64);
    rs = rsqrtf(ssq / (float)K + p.eps);
  }
  float py = 0.f, px = 0.f;
  if (EPI == KD_EPI_QKV) {
    const int tok = rowc % p.rows_per_sample;
    py = p.pos[2 * tok];
    px = p.pos[2 * tok + 1];
  }
  code_warm_end(warm);
  // the E8M0 channel-scale bytes of this workgroup's n-tiles -> LDS (one dword per (tile, l31): the four feature groups' bytes)
  unsigned* wsl = reinterpret_cast<unsigned*>(smem + NSTG * WBLK + NWV * K * 4);
  for (int i = tid; i < n_tiles * 32; i += 256) wsl[i] = p.Ws[(size_t)nt_begin * 32 + i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");              // every wave has taken its rows out of the slot it borrowed; the scale bytes are in
#pragma unroll
  for (int s = 0; s < PDIST; ++s)
    if (s < total) issue(s);
  const bool full_panel = m0 + 128 <= p.M;
  if (probe) p.clk[4] = __builtin_amdgcn_s_memtime();
  u16* crow = p.C + (size_t)rowc * p.N;
  const int rd = lh * 512 + l31 * 16;                  // this lane's 16 bytes inside a (j, ks, h) KiB of a block

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int nt = 0; nt < n_tiles; ++nt) {
    // the tile's four E8M0 channel-scale bytes of this lane's W rows (feature groups j = 0..3), each spread over a register's byte lanes
    const unsigned wsw = wsl[nt * 32 + l31];
    int wsc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wsc[j] = (int)(((wsw >> (8 * j)) & 0xFFu) * 0x01010101u);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int s = nt * NKB + kb;
      {
        int allow = PB * min(PDIST - 1, total - 1 - s);
        if (full_panel) {
          if (nt > 0 && kb + 1 <= PDIST) allow += NST;
          if (nt > 1 && kb + 1 + NKB <= PDIST) allow += NST;
        }
        wait_vm_dyn8(allow);
      }
      asm volatile("s_barrier" ::: "memory");
      if (s + PDIST < total) issue(s + PDIST);
      const char* st = smem + (s % NSTG) * WBLK + rd;
      i32x8 wf[2][4];
      auto read4 = [&](int ks, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 h0 = *reinterpret_cast<const u32x4*>(st + ((j * 2 + ks) * 2 + 0) * 1024);
          const u32x4 h1 = *reinterpret_cast<const u32x4*>(st + ((j * 2 + ks) * 2 + 1) * 1024);
          wf[buf][j] = i32x8{(int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
        }
      };
      read4(0, 0);
      read4(1, 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[ks][j], a[2 * kb + ks], acc[j], 0, 0, 0, wsc[j], 0, asc[2 * kb + ks]);
      }
    }
    const int n0 = (nt_begin + nt) * NCOL;
    if (probe && nt == 0) p.clk[5] = __builtin_amdgcn_s_memtime();
    if (GEGLU) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float v[16];
        const float rsh = 0.5f * rs;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 o = geglu_pair(f32x2{acc[2 * jj][r], acc[2 * jj][r + 1]} * rsh, f32x2{acc[2 * jj + 1][r], acc[2 * jj + 1][r + 1]} * rs);
          v[r] = o.x;
          v[r + 1] = o.y;
        }
        if constexpr (C8OUT) {
          // v[4 g + e] = feature 8 g + 4 lh + e of the block: the block's maximum over both lanes of the row, then 16 e4m3 bytes per lane --
          // after the half-wave exchange lane half 0 holds features 0..15, lane half 1 features 16..31: one 16-byte store each
          float amax = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(v[r]));
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
          const unsigned sb = mx_scale_byte(amax);
          const float inv = mx_inv_scale(sb);
          unsigned w[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int pk = 0;
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g] * inv, v[4 * g + 1] * inv, pk, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * g + 2] * inv, v[4 * g + 3] * inv, pk, true);
            w[g] = (unsigned)pk;
          }
          half_swap(w[0], w[2]);
          half_swap(w[1], w[3]);
          if (ok) {
            st16(p.C8 + (size_t)rowc * p.N + n0 + 32 * jj + 16 * lh, u32x4{w[0], w[2], w[1], w[3]});
            if (lh == 0) p.Cs[(size_t)rowc * (p.N >> 5) + ((n0 + 32 * jj) >> 5)] = (unsigned char)sb;
          }
        } else {
          store_block_bf16(crow + n0 + 32 * jj, v, lh, ok);
        }
      }
    } else if (EPI == KD_EPI_QKV) {
#pragma unroll
      for (int vv = 0; vv < 2; ++vv) {
        const int vec = (n0 >> 6) + vv;
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
          qk_prep_blocks(acc[2 * vv], acc[2 * vv + 1], rs, sqrtf(qsc), p.eps, py, px, fr);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[2 * vv][r] *= rs; acc[2 * vv + 1][r] *= rs; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[2 * vv + jj][r];
          store_block_bf16(crow + n0 + 64 * vv + 32 * jj, v, lh, ok);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[j][r] * rs;
        store_block_bf16(crow + n0 + 32 * j, v, lh, ok);
      }
    }
    if (probe && nt == 0) p.clk[6] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  if (probe) { p.clk[2] = __builtin_amdgcn_s_memtime(); p.clk[3] = __builtin_amdgcn_s_memrealtime(); p.clk[7] = (unsigned long long)n_tiles; }
}

// ---- tiled form: BOTH operands e4m3 through LDS (down projection + residual of the fp8 mode) ------------------------------------------------
// gemm_bf16.hip's tiled kernel (128 x 128 tile per workgroup, 2 x 2 blocks per wave, both operands by LDS-DMA, K loop = ds_read + matrix
// instruction) with A = the e4m3 rows + E8M0 block scales a C8OUT launch above wrote and W = a kd_pack_weight_mx8 image.  A ring slot covers
// 128 k: A image [128 rows][128 bytes] (16-byte chunk q of row r at q ^ ((r >> 1) & 7), permuted on the source side: bf16_common.h swz128)
// + one 16 KiB W block = 32 KiB, 8 instructions per wave.  The row's scale bytes (K / 32) are read once into registers; the byte of the
// block a lane feeds (block lh of the k-step) is spread over the register's byte lanes by one v_perm_b32.
struct TArgs8 {
  const unsigned char* A8; const unsigned* As; const char* Wp; const unsigned* Ws; u16* C; const u16* R;
  int M, N, K, n_tiles_n;
  int warm;
};

template <int NKB /* K / 128 */, int EPI, bool DEEP>
__global__ __launch_bounds__(256, DEEP ? 1 : 2) void gemm_mx8_tiled_kernel(const TArgs8 p) {
  constexpr int SLOT = 2 * WBLK, NSTG = DEEP ? 4 : 2, K = NKB * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const auto warm = code_warm_begin<8192>((int)blockIdx.x < p.warm && tid < 64);
  const int wc = wid & 1, wr = wid >> 1;
  int tile;
  {   // XCD-aware order, n fastest: the n-tiles of one row panel run back to back on ONE L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nt = tile % p.n_tiles_n, mt = tile / p.n_tiles_n;
  const int m0 = mt * 128, n0 = nt * 128;
  const char* aptr[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int row = 8 * (4 * wid + ii) + (lane >> 3);
    const int q = (lane & 7) ^ ((row >> 1) & 7);
    aptr[ii] = reinterpret_cast<const char*>(p.A8) + (size_t)min(m0 + row, p.M - 1) * K + q * 16;
  }
  const char* wsrc = p.Wp + (size_t)nt * NKB * WBLK + (wid * 4) * 1024 + lane * 16;
  auto issue = [&](int kt) {
    char* st = smem + (kt % NSTG) * SLOT;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) glds16(aptr[ii] + kt * 128, st + (4 * wid + ii) * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(wsrc + (size_t)kt * WBLK + j * 1024, st + WBLK + (wid * 4 + j) * 1024);
  };
#pragma unroll
  for (int kt = 0; kt < NSTG - 1; ++kt)
    if (kt < NKB) issue(kt);
  // this lane's rows of the two row blocks: their scale bytes (4 per 128 k), and the channel-scale bytes of the wave's two feature groups
  unsigned asw[2][NKB];
  int wsc[2];
  {
    const unsigned wsw = p.Ws[(size_t)nt * 32 + l31];
#pragma unroll
    for (int i = 0; i < 2; ++i) wsc[i] = (int)(((wsw >> (8 * (2 * wc + i))) & 0xFFu) * 0x01010101u);
#pragma unroll
    for (int jr = 0; jr < 2; ++jr) {
      const unsigned* sp = p.As + (size_t)min(m0 + wr * 64 + 32 * jr + l31, p.M - 1) * NKB;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) asw[jr][kb] = sp[kb];
    }
  }
  const unsigned sel0 = (unsigned)lh * 0x01010101u, sel1 = (unsigned)(2 + lh) * 0x01010101u;      // v_perm_b32 selectors: byte lh / 2 + lh of a dword to all four lanes
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr bool HAS_R = EPI == KD_EPI_RESIDUAL;
  u32x4 rraw[2][2][2];
  size_t roff[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      roff[i][j] = (size_t)min(m0 + wr * 64 + 32 * j + l31, p.M - 1) * p.N + min(n0 + wc * 64 + 32 * i, p.N - 32);
  code_warm_end(warm);
#pragma unroll
  for (int kt = 0; kt < NKB; ++kt) {
    // the steps requested after kt may stay in flight (8 LDS-DMA requests per wave and step); the scale loads above were consumed by the compiler's own wait
    if (NSTG == 2 || kt + 1 >= NKB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (kt + 2 >= NKB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (kt + NSTG - 1 < NKB) issue(kt + NSTG - 1);
    if (HAS_R && kt == (NKB >= 2 ? NKB - 2 : 0)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) load_block_raw(p.R + roff[i][j], rraw[i][j], lh);
    }
    const char* st = smem + (kt % NSTG) * SLOT;
    const char* wb = st + WBLK + lh * 512 + l31 * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i32x8 af[2], wf[2];
#pragma unroll
      for (int jr = 0; jr < 2; ++jr) {
        const int row = wr * 64 + 32 * jr + l31;
        const u32x4 h0 = *reinterpret_cast<const u32x4*>(st + swz128(row, 4 * ks + lh));
        const u32x4 h1 = *reinterpret_cast<const u32x4*>(st + swz128(row, 4 * ks + 2 + lh));
        af[jr] = i32x8{(int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = 2 * wc + i;
        const u32x4 h0 = *reinterpret_cast<const u32x4*>(wb + ((j * 2 + ks) * 2 + 0) * 1024);
        const u32x4 h1 = *reinterpret_cast<const u32x4*>(wb + ((j * 2 + ks) * 2 + 1) * 1024);
        wf[i] = i32x8{(int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
      }
#pragma unroll
      for (int jr = 0; jr < 2; ++jr) {
        const int sc = (int)__builtin_amdgcn_perm(asw[jr][kt], asw[jr][kt], ks ? sel1 : sel0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][jr] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[jr], acc[i][jr], 0, 0, 0, wsc[i], 0, sc);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gm = m0 + wr * 64 + 32 * j + l31;
    const bool ok = gm < p.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = n0 + wc * 64 + 32 * i;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
      if (HAS_R) {
        float rr_[16];
        block_from_raw(rraw[i][j], rr_);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rr_[r];
      }
      store_block_bf16(p.C + roff[i][j], v, lh, ok && nb < p.N);
    }
  }
}

// ---- one-off weight packing: fp32 W [N, K] (GEGLU: [2 N, K]) -> e4m3 blocks + E8M0 channel scales -----------------------------------------
// One wave per tile row: amax over K, the power-of-two channel scale (checkpoint.quantize_fp8's rule), RNE conversion.
__global__ __launch_bounds__(64) void pack_weight_mx8_kernel(const float* __restrict__ W, char* __restrict__ out, int N, int K, int geglu, int n_tiles) {
  const int tr = blockIdx.x, nt = tr >> 7, r = tr & 127, lane = threadIdx.x;
  const int src = w_row_of_tile(nt, r, N, geglu != 0);
  const int nkb = K / 128;
  float amax = 0.f;
  if (src >= 0)
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(W[(size_t)src * K + k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const unsigned sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  const int j = r >> 5, l31 = r & 31;
  for (int k4 = lane * 4; k4 < K; k4 += 256) {          // 4 consecutive k = one dword of a lane's fragment
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = W[(size_t)src * K + k4 + e] * inv;
    }
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
    const int kb = k4 >> 7, ks = (k4 >> 6) & 1, h = (k4 >> 5) & 1, lh = (k4 >> 4) & 1, byte = k4 & 15;      // h = 32-k block of the step, lh = its half
    const size_t off = ((size_t)nt * nkb + kb) * WBLK + ((((j * 2 + ks) * 2 + h) * 2 + lh) * 32 + l31) * 16 + byte;
    *reinterpret_cast<int*>(out + off) = pk;
  }
  if (lane == 0) out[(size_t)n_tiles * nkb * WBLK + (size_t)nt * 128 + l31 * 4 + j] = (char)sb;
}

}  // namespace mx8
}  // namespace kd

using namespace kd;
using namespace kd::mx8;

static int mx8_tiles(int N, int geglu) { return geglu ? (N + 63) / 64 : (N + 127) / 128; }

extern "C" long long kd_packed_weight_bytes_mx8(int N, int K, int geglu) {
  if (N <= 0 || K <= 0 || K % 128) return -1;
  const long long nt = mx8_tiles(N, geglu);
  return nt * (K / 128) * WBLK + nt * 128;
}

extern "C" int kd_pack_weight_mx8(const float* W, void* out, int N, int K, int geglu, void* stream) {
  if (!W || !out) return fail(KD_EINVAL, "kd_pack_weight_mx8: null operand");
  if (N <= 0 || K <= 0 || K % 128) return fail(KD_EINVAL, "kd_pack_weight_mx8: N=%d K=%d (K must be a multiple of 128)", N, K);
  const int nt = mx8_tiles(N, geglu);
  hipLaunchKernelGGL(pack_weight_mx8_kernel, dim3((unsigned)(nt * 128)), dim3(64), 0, (hipStream_t)stream, W, reinterpret_cast<char*>(out), N, K, geglu, nt);
  return check_launch("kd_pack_weight_mx8");
}

// Products on the block-scaled fp8 matrix instruction.  `d` is the projection's descriptor as kd_gemm_bf16 takes it (bf16 activations,
// precision = KD_PREC_BF16) except that Wp points at the kd_pack_weight_mx8 image.  Two forms:
//   norm = 1 (AdaRMSNorm -> projection; epi = KD_EPI_STORE / KD_EPI_QKV / KD_EPI_GEGLU; K in {256, 512}, N a multiple of the tile (128; GEGLU:
//            64), M >= 128): the A-stationary kernel; the activations are quantised in the kernel.  With c_split = 1 (GEGLU only) the
//            result leaves as e4m3 rows in C ([M, N] bytes) + one E8M0 byte per (row, 32 features) in C_lo ([M, N / 32] bytes);
//   norm = 0, a_split = 1 (epi = KD_EPI_STORE / KD_EPI_RESIDUAL; K in {256, 512, 768, 1536}, N a multiple of 128): A is such an e4m3 tensor
//            (A = [M, K] bytes, A_lo = [M, K / 32] scale bytes): the tiled kernel, both operands by LDS-DMA (the fp8 mode's down projection).
extern "C" int kd_gemm_mx8_supported(int M, int N, int K, int epi, int norm) {
  if (M < 128 || !option("mx8", 1)) return 0;
  if (!norm) return (epi == KD_EPI_STORE || epi == KD_EPI_RESIDUAL) && (K == 256 || K == 512 || K == 768 || K == 1536) && N > 0 && N % 128 == 0;
  if (K != 256 && K != 512) return 0;
  if (epi == KD_EPI_GEGLU) return N > 0 && N % 64 == 0;
  if (epi == KD_EPI_QKV) return N == 3 * K;
  if (epi == KD_EPI_STORE) return N > 0 && N % 128 == 0;
  return 0;
}

static int mx8_tiled(const KdGemm& d, hipStream_t s) {
  if (!d.A || !d.A_lo || !d.Wp || !d.C || (d.epi == KD_EPI_RESIDUAL && !d.R)) return fail(KD_EINVAL, "kd_gemm_mx8: null operand");
  if ((reinterpret_cast<size_t>(d.A) & 15) || (reinterpret_cast<size_t>(d.A_lo) & 3))
    return fail(KD_EINVAL, "kd_gemm_mx8: the e4m3 rows must be 16-byte aligned and their scale bytes 4-byte aligned");
  if (d.epi == KD_EPI_STORE && d.out_add != 0.f) return fail(KD_EINVAL, "kd_gemm_mx8: out_add is not taken");
  const int n_tiles_n = d.N / 128, nkb = d.K / 128;
  TArgs8 a{reinterpret_cast<const unsigned char*>(d.A), reinterpret_cast<const unsigned*>(d.A_lo), reinterpret_cast<const char*>(d.Wp),
           reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(d.Wp) + (size_t)n_tiles_n * nkb * WBLK), reinterpret_cast<u16*>(d.C),
           reinterpret_cast<const u16*>(d.R), d.M, d.N, d.K, n_tiles_n, option("code_warm", KD_CODE_WARM_DEFAULT)};
  const long tiles = (long)((d.M + 127) / 128) * n_tiles_n;
  const bool deep = tiles <= cu_count();              // at most one tile per CU: a 4-slot ring instead of two workgroups per CU
  const double flops = 2.0 * d.M * (double)d.N * d.K;
  const double bytes = (double)d.M * d.K * (1.0 + 1.0 / 32) + (double)d.N * d.K + 2.0 * d.M * d.N * (d.epi == KD_EPI_RESIDUAL ? 2.0 : 1.0);
  char nm[96] = "gemm_mx8_tiled";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_mx8_tiled<e%d> M=%d N=%d K=%d", d.epi, d.M, d.N, d.K);
  LaunchScope prof(nm, flops, bytes, s);
#define KD_MT(NKBV, EP, DP) { constexpr int LDS = (DP ? 4 : 2) * 2 * WBLK; static LdsAttr set;                                \
    set.ensure(reinterpret_cast<const void*>(gemm_mx8_tiled_kernel<NKBV, EP, DP>), LDS);                                          \
    hipLaunchKernelGGL((gemm_mx8_tiled_kernel<NKBV, EP, DP>), dim3((unsigned)tiles), dim3(256), LDS, s, a); }
#define KD_MT2(NKBV) { if (d.epi == KD_EPI_RESIDUAL) { if (deep) KD_MT(NKBV, KD_EPI_RESIDUAL, true) else KD_MT(NKBV, KD_EPI_RESIDUAL, false) } \
                       else { if (deep) KD_MT(NKBV, KD_EPI_STORE, true) else KD_MT(NKBV, KD_EPI_STORE, false) } }
  if (nkb == 2) KD_MT2(2) else if (nkb == 4) KD_MT2(4) else if (nkb == 6) KD_MT2(6) else KD_MT2(12)
#undef KD_MT2
#undef KD_MT
  return check_launch("kd_gemm_mx8(tiled)");
}

extern "C" int kd_gemm_mx8(const KdGemm* dp, void* stream) {
  if (!dp) return fail(KD_EINVAL, "kd_gemm_mx8: null descriptor");
  const KdGemm& d = *dp;
  if (d.a_mode != KD_A_PLAIN || d.precision != KD_PREC_BF16)
    return fail(KD_EINVAL, "kd_gemm_mx8: the descriptor must be a bf16-mode projection with a plain A operand");
  if (!kd_gemm_mx8_supported(d.M, d.N, d.K, d.epi, d.norm))
    return fail(KD_EINVAL, "kd_gemm_mx8: shape M=%d N=%d K=%d epi=%d norm=%d is not taken", d.M, d.N, d.K, d.epi, d.norm);
  if (!d.norm) {
    if (!d.a_split) return fail(KD_EINVAL, "kd_gemm_mx8: without a norm the A operand must be e4m3 rows + block scales (a_split = 1, A_lo = the scale bytes)");
    return mx8_tiled(d, (hipStream_t)stream);
  }
  if (!d.A || !d.Wp || !d.C || !d.scale) return fail(KD_EINVAL, "kd_gemm_mx8: null operand");
  if (d.a_split || d.rows_per_sample <= 0 || (d.epi == KD_EPI_STORE && d.out_add != 0.f))
    return fail(KD_EINVAL, "kd_gemm_mx8: the descriptor must be a bf16 norm -> projection (bf16 A, rows_per_sample set)");
  if (d.c_split && (d.epi != KD_EPI_GEGLU || !d.C_lo)) return fail(KD_EINVAL, "kd_gemm_mx8: the e4m3 output (c_split) is the GEGLU epilogue's, with C_lo = the scale bytes");
  if (d.epi == KD_EPI_QKV && (!d.qk_scale || !d.rope_pos || !d.rope_freq || d.n_heads * 64 != d.K))
    return fail(KD_EINVAL, "kd_gemm_mx8: the qkv projection needs qk_scale, rope_pos, rope_freq and n_heads * 64 == K");
  const bool geglu = d.epi == KD_EPI_GEGLU;
  const int n_tiles = mx8_tiles(d.N, geglu), nkb = d.K / 128;
  MArgs a{reinterpret_cast<const u16*>(d.A), reinterpret_cast<const char*>(d.Wp),
          reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(d.Wp) + (size_t)n_tiles * nkb * WBLK), reinterpret_cast<u16*>(d.C),
          reinterpret_cast<unsigned char*>(d.C), reinterpret_cast<unsigned char*>(d.C_lo),
          d.scale, d.scale_stride, d.rows_per_sample, d.eps, d.M, d.N, n_tiles, 1,
          d.n_heads, d.qk_scale, d.rope_pos, d.rope_freq, option("code_warm", KD_CODE_WARM_DEFAULT), g_clk};
  // n-splits of a panel: gemm_astat's cost model (rounds x (row prologue + tiles per split)), two workgroups per CU
  const int panels = (d.M + 127) / 128, slots = 2 * cu_count();
  int best = 1;
  long best_cost = -1;
  for (int sp = 1; sp <= n_tiles; ++sp) {
    if (n_tiles % sp) continue;
    const long rounds = ((long)panels * sp + slots - 1) / slots;
    const long cost = rounds * (2 + n_tiles / sp);
    if (best_cost < 0 || cost < best_cost) { best = sp; best_cost = cost; }
  }
  const int forced = option("mx8_splits", 0);
  a.n_splits = forced > 0 && forced <= n_tiles ? forced : best;
  if ((n_tiles + a.n_splits - 1) / a.n_splits > 64) return fail(KD_EINVAL, "kd_gemm_mx8: more than 64 n-tiles per workgroup (N=%d)", d.N);
  hipStream_t s = (hipStream_t)stream;
  const double n_eff = geglu ? 2.0 * d.N : (double)d.N;
  const double flops = 2.0 * d.M * n_eff * d.K;
  const double bytes = 2.0 * (double)d.M * d.K + (d.c_split ? (1.0 + 1.0 / 32) : 2.0) * (double)d.M * d.N + n_eff * d.K;
  char nm[96] = "gemm_mx8_astat";
  if (prof_on()) snprintf(nm, sizeof(nm), "gemm_mx8_astat<e%d%s> M=%d N=%d K=%d", d.epi, d.c_split ? ",c8" : "", d.M, d.N, d.K);
  LaunchScope prof(nm, flops, bytes, s);
  // ring + one scale vector (K floats) per wave + the channel-scale bytes of a split's n-tiles (<= 64 tiles)
#define KD_MX(NKV, EP, C8) { constexpr int LDS = 4 * WBLK + 4 * NKV * 64 * 4 + 64 * 128; static LdsAttr set;           \
    set.ensure(reinterpret_cast<const void*>(gemm_mx8_astat_kernel<NKV, EP, C8>), LDS);                                  \
    hipLaunchKernelGGL((gemm_mx8_astat_kernel<NKV, EP, C8>), dim3((unsigned)(panels * a.n_splits)), dim3(256), LDS, s, a); }
  if (d.K == 512) {
    if (d.epi == KD_EPI_GEGLU) { if (d.c_split) KD_MX(8, KD_EPI_GEGLU, true) else KD_MX(8, KD_EPI_GEGLU, false) }
    else if (d.epi == KD_EPI_QKV) KD_MX(8, KD_EPI_QKV, false) else KD_MX(8, KD_EPI_STORE, false)
  } else {
    if (d.epi == KD_EPI_GEGLU) { if (d.c_split) KD_MX(4, KD_EPI_GEGLU, true) else KD_MX(4, KD_EPI_GEGLU, false) }
    else if (d.epi == KD_EPI_QKV) KD_MX(4, KD_EPI_QKV, false) else KD_MX(4, KD_EPI_STORE, false)
  }
#undef KD_MX
  return check_launch("kd_gemm_mx8");
}

KD_TEXT_PAD(gemm_mx8)      // last function of this code object: kd_common.h, code warm-up
