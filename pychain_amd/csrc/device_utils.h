// device_utils.h - small device helpers shared by the den and num kernels.
#ifndef PYCHAIN_HIP_DEVICE_UTILS_H_
#define PYCHAIN_HIP_DEVICE_UTILS_H_

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace pychain_hip {

// ---- LDS by absolute byte address (so an operand address is ONE VGPR, no base add) ----
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(lds_char*)(p);
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
__device__ __forceinline__ float lds_abs(uint32_t byte_addr) { return *(lds_cfloat*)(byte_addr); }
#pragma clang diagnostic pop

// A sequence length as the kernels use it: clamped to [1, T].  Lengths that live on the device are never seen
// by the host-side validation (that would be a sync); an out-of-range one must not turn into out-of-bounds
// reads and writes.  The recursion kernels also count it into `bad` (seq_len_bad).
__device__ __forceinline__ int seq_len(const int64_t* lengths, int b, int T) {
  const int64_t l = lengths[b];
  return l < 1 ? 1 : (l > T ? T : (int)l);
}
__device__ __forceinline__ bool seq_len_bad(const int64_t* lengths, int b, int T) {
  const int64_t l = lengths[b];
  return l < 1 || l > T;
}

// ---- wave64 reductions on DPP (no LDS traffic, unlike __shfl_xor = ds_bpermute) --------
#define PYCHAIN_DPP_ADD(v, ctrl) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true)))
// sum over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float dpp_row_sum(float v) {
  v = PYCHAIN_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
  v = PYCHAIN_DPP_ADD(v, 0x4E);    // quad_perm [2,3,0,1]
  v = PYCHAIN_DPP_ADD(v, 0x141);   // row_half_mirror
  v = PYCHAIN_DPP_ADD(v, 0x140);   // row_mirror
  return v;
}
// sum over the wave, same value (an SGPR-derived one) in every lane
// (the four row sums r0 .. r3 are combined by two more DPP steps - row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3:
// lane 63 then holds (r3 + r2) + (r1 + r0), bit for bit the (r0 + r1) + (r2 + r3) of four readlanes and three adds - and ONE
// readlane: 7 VALU instructions instead of 13.  The recursion frames sit on their VALU issue - DESIGN.md 3.16 - and every wave
// reduces one or two such sums per frame.)
#ifndef PYCHAIN_WAVE_SUM_BCAST
#define PYCHAIN_WAVE_SUM_BCAST 1
#endif
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_row_sum(v);
#if PYCHAIN_WAVE_SUM_BCAST
  // (written out: from the builtin the compiler makes v_mov 0, v_mov_dpp, v_add of each step; the s_nop is the two wait states a
  // DPP read needs after the VALU write of its source, which the compiler cannot see into the asm to insert)
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));   // rows 1, 3 += lane 15 of rows 0, 2
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));   // rows 2, 3 += lane 31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
#endif
}

// exp(c) for |c| <= 30.  PYCHAIN_EXP_OPS == 5 (rounds 1-2): v_exp_f32 on c*log2(e) with the rounding error of that product
// fed back, exp(c) = 2^t * (1 + r ln2): relative error ~2e-7 whatever |c|.  PYCHAIN_EXP_OPS == 2 (round 3, default): v_exp_f32
// on the rounded product alone: the error of t = c*log2(e) in fp32 is |t| * 6e-8, i.e. a relative error of |c| * 6e-8 in the
// result - 2e-7 at |c| = 3 (network outputs are O(1)), 1.8e-6 at the clamp (|c| = 30) - against a parity bar of 1e-4 and an
// oracle whose own expf is 1 ulp.  The clamp / exp of a row sits in the serial tail of every recursion frame, where the four
// waves of a SIMD issue it one after another: three instructions less per element are ~150 cycles per frame.
#ifndef PYCHAIN_EXP_OPS
#define PYCHAIN_EXP_OPS 2
#endif
__device__ __forceinline__ float exp_bounded(float c) {
  const float kL2E = 1.44269502162933349609375f;       // fp32(log2 e)
  const float t = c * kL2E;
#if PYCHAIN_EXP_OPS == 2
  return __builtin_amdgcn_exp2f(t);
#else
  const float r = fmaf(c, kL2E, -t);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147182464599609375f, e);
#endif
}

// How a raw nnet-output element enters LDS (pychain/loss.py:30,43):
//   kXExpClamp  exp(clamp(v,-30,30))   denominator fed raw network output (clamp+exp fused)
//   kXIdentity  v                      denominator fed exp'd input (pychain_C.forward_backward contract)
//   kXClamp     clamp(v,-30,30)        numerator (log domain)
enum { kXExpClamp = 0, kXIdentity = 1, kXClamp = 2 };
// (v_med3_f32 turns a NaN into -30: the rows are watched for NaN separately, XRow::has_nan, by the one
// workgroup per sequence that reports its log-probability - a select per element here costs the recursions
// 4 % because the row's VALU work sits on their critical path.)
__device__ __forceinline__ float clamp_exp(float v, int mode) {
  if (mode == kXIdentity) return v;
  const float c = __builtin_amdgcn_fmed3f(v, -30.f, 30.f);
  return mode == kXClamp ? c : exp_bounded(c);
}

// a*b + c*d and a*b + c with every product ROUNDED before the sum (no fma contraction: HIP compiles
// with -ffp-contract=fast and __fmul_rn/__fadd_rn are plain operators there).  Used where two code
// paths must produce the same bits (gradient written once vs. written, then accumulated into).
__device__ __forceinline__ float mul_add_mul_rn(float a, float b, float c, float d) {
#pragma clang fp contract(off)
  const float p = a * b;
  const float q = c * d;
  return p + q;
}
// g = a*b rounded, sum += g rounded: the one-frame and the two-frame occupancy kernel build their frame
// totals from the same partial sums, and must not differ by where the compiler happens to form an fma
__device__ __forceinline__ float product_into_sum(float a, float b, float& sum) {
#pragma clang fp contract(off)
  const float g = a * b;
  sum = sum + g;
  return g;
}
__device__ __forceinline__ float mul_add_rn(float a, float b, float c) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p + c;
}

// ---- 2-byte network outputs (bf16 / fp16: SURVEY.md row f4, DenArgs::x_half) -----------------------------------------------
// The kernels that read the [B,T,D] network output take it as it is - 2-byte elements converted where they land, the
// gradient rounded to the same type where it is written - instead of a host-side up-cast pass, a second [B,T,D] fp32
// buffer and a cast of the gradient back (pychain_amd/native.py).  fmt: 1 = bf16, 2 = fp16 (uniform per call).
enum { kXF32 = 0, kXBf16 = 1, kXF16 = 2 };
__device__ __forceinline__ float half_bits_to_f32(uint32_t h /* low 16 bits */, bool bf16) {
  if (bf16) return __uint_as_float(h << 16);
  return __half2float(__ushort_as_half((unsigned short)h));
}
// the two elements of one dword (element 2i in the low half)
__device__ __forceinline__ void half2_to_f32(uint32_t w, bool bf16, float& lo, float& hi) {
  if (bf16) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
  else { lo = __half2float(__ushort_as_half((unsigned short)(w & 0xffffu))); hi = __half2float(__ushort_as_half((unsigned short)(w >> 16))); }
}
// round to nearest even, as torch's .to(dtype) does (a NaN stays a NaN)
__device__ __forceinline__ uint32_t f32_to_half_bits(float f, bool bf16) {
  if (bf16) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  return (uint32_t)__half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ uint32_t pack_half2(float lo, float hi, bool bf16) {
  return f32_to_half_bits(lo, bf16) | (f32_to_half_bits(hi, bf16) << 16);
}

// ---- buffer descriptor of one sequence's nnet-output [T, D] -------------------------------------
// Rows are fetched with `buffer_load_dwordx4 vdst, voffset, srsrc, soffset offen`: the row is selected by
// the SGPR soffset, the lane by a loop-invariant 32-bit voffset, so a frame's load writes NO address
// VGPR.  (With a 64-bit VGPR address the per-frame address arithmetic overwrites registers an
// outstanding store may still read, and the compiler drains every VMEM operation - the previous
// frame's row store to HBM included - before the frame can start.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t XBuf;
__device__ __forceinline__ XBuf make_xbuf(const float* seq, size_t bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(seq);     // uniformity made provable: both halves through readfirstlane
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffull ? 0xffffffffull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}

// ---- nnet-output row: global -> registers (early) -> LDS (late) ------------------
template <int NT, int VEC, int XCH>
struct XRow {
  float v[(VEC * XCH) > 0 ? (VEC * XCH) : 1];
  // row t of the sequence behind `buf` (VEC == 4 only): see make_xbuf
  __device__ __forceinline__ void load_row(XBuf buf, int t, int D, int tid) {
    static_assert(VEC == 4 && XCH > 0, "buffer form: float4 chunks");
    const int soff = __builtin_amdgcn_readfirstlane(t * D * 4);
#pragma unroll
    for (int c = 0; c < XCH; c++) {
      const int e = (c * NT + tid) * 4;
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(buf, min(e, D - 4) * 4, soff, 0);   // lanes past the row re-read its end
      v[c * 4 + 0] = __uint_as_float(q.x); v[c * 4 + 1] = __uint_as_float(q.y);
      v[c * 4 + 2] = __uint_as_float(q.z); v[c * 4 + 3] = __uint_as_float(q.w);
    }
  }
  // The same two loads for 2-byte rows (VEC == 4: four elements = 8 bytes per thread and chunk): `row` / `buf` address 2-byte
  // elements.  The RAW words stay in v[4c], v[4c + 1] as loaded - converting them on the spot would make the wave wait for
  // the data where it asked for it, and every caller asks a frame ahead of the use - and convert_h() expands them in place
  // right before the values are used (exactly once per load).
  __device__ __forceinline__ void load_row_h(XBuf buf, int t, int D, int tid) {
    static_assert(VEC == 4 && XCH > 0, "buffer form: chunks of four elements");
    const int soff = __builtin_amdgcn_readfirstlane(t * D * 2);
#pragma unroll
    for (int c = 0; c < XCH; c++) {
      const int e = (c * NT + tid) * 4;
      typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
      const u32x2_ q = __builtin_amdgcn_raw_buffer_load_b64(buf, min(e, D - 4) * 2, soff, 0);
      v[c * 4 + 0] = __uint_as_float(q.x); v[c * 4 + 1] = __uint_as_float(q.y);
    }
  }
  __device__ __forceinline__ void load_h(const void* __restrict__ row, int D, int tid) {
    static_assert(VEC == 4 && XCH > 0, "2-byte rows: chunks of four elements (D % 4 == 0)");
#pragma unroll
    for (int c = 0; c < XCH; c++) {
      const int e = (c * NT + tid) * 4;
      const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(row) + (size_t)min(e, D - 4) * 2);
      v[c * 4 + 0] = __uint_as_float(q.x); v[c * 4 + 1] = __uint_as_float(q.y);
    }
  }
  // (`bf16` is uniform: ONE branch around the whole row, not one per pair of elements - sixteen branch sites in a staging
  // loop whose pace is its instruction stream made the numerator's step a third longer)
  template <bool BF16>
  __device__ __forceinline__ void convert_h_as() {
#pragma unroll
    for (int c = 0; c < XCH; c++) {
      const uint32_t w0 = __float_as_uint(v[c * 4 + 0]), w1 = __float_as_uint(v[c * 4 + 1]);
      half2_to_f32(w0, BF16, v[c * 4 + 0], v[c * 4 + 1]);
      half2_to_f32(w1, BF16, v[c * 4 + 2], v[c * 4 + 3]);
    }
  }
  __device__ __forceinline__ void convert_h(bool bf16) {
    static_assert(VEC == 4 && XCH > 0, "2-byte rows: chunks of four elements");
    if (bf16) convert_h_as<true>(); else convert_h_as<false>();
  }
  __device__ __forceinline__ void load(const float* __restrict__ row, int D, int tid) {
    if constexpr (XCH > 0) {
#pragma unroll
      for (int c = 0; c < XCH; c++) {
        const int e = (c * NT + tid) * VEC;
        if constexpr (VEC == 4) {
          // unpredicated: lanes past the row re-read its last float4 (store() never uses them)
          const float4 q = *reinterpret_cast<const float4*>(row + min(e, D - 4));
          v[c * 4 + 0] = q.x; v[c * 4 + 1] = q.y; v[c * 4 + 2] = q.z; v[c * 4 + 3] = q.w;
        } else {
          v[c] = e < D ? row[e] : 0.f;
        }
      }
    }
  }
  // true if one of the staged elements is NaN (torch.clamp / exp propagate it to the loss, loss.py:30,43;
  // the fused clamp would hide it).  One unordered compare per PAIR of elements.
  __device__ __forceinline__ bool has_nan() const {
    bool n = false;
    if constexpr (XCH > 0) {
      constexpr int N = VEC * XCH;
#pragma unroll
      for (int i = 0; i + 1 < N; i += 2) n = n || __builtin_isunordered(v[i], v[i + 1]);
      if (N & 1) n = n || v[N - 1] != v[N - 1];
    }
    return n;
  }
  // `mode` is uniform: one branch per call, not a select per element
  template <int MODE>
  __device__ __forceinline__ void store_mode(float* lds, int D, int tid) {
#pragma unroll
    for (int c = 0; c < XCH; c++) {
      const int e = (c * NT + tid) * VEC;
      if (e < D) {
        if constexpr (VEC == 4) {
          float4 q;
          q.x = clamp_exp(v[c * 4 + 0], MODE); q.y = clamp_exp(v[c * 4 + 1], MODE);
          q.z = clamp_exp(v[c * 4 + 2], MODE); q.w = clamp_exp(v[c * 4 + 3], MODE);
          *reinterpret_cast<float4*>(lds + e) = q;
        } else {
          lds[e] = clamp_exp(v[c], MODE);
        }
      }
    }
  }
  // returns true if the register-less form (XCH == 0) met a NaN element (the register forms are asked with has_nan())
  __device__ __forceinline__ bool store(float* lds, const float* __restrict__ row, int D, int tid, int is_exp) {
    if constexpr (XCH > 0) {
      if (is_exp == kXExpClamp) store_mode<kXExpClamp>(lds, D, tid);
      else if (is_exp == kXIdentity) store_mode<kXIdentity>(lds, D, tid);
      else {
        store_mode<kXClamp>(lds, D, tid);
        // numerator rows: a NaN stays a NaN in LDS (it reaches the log-probability only if an arc of the
        // utterance's graph emits that pdf, as in the reference).  Rare path: the common one pays one compare
        // per pair of elements, not a select per element.
        if (has_nan()) {
#pragma unroll
          for (int c = 0; c < XCH; c++)
#pragma unroll
            for (int k = 0; k < VEC; k++) {
              const int e = (c * NT + tid) * VEC + k;
              if (e < D && v[c * VEC + k] != v[c * VEC + k]) lds[e] = v[c * VEC + k];
            }
        }
      }
      return false;
    } else {  // any D: no register staging
      bool nan = false;
      for (int e = tid; e < D; e += NT) {
        const float r = row[e];
        nan = nan || r != r;
        lds[e] = (is_exp == kXClamp && r != r) ? r : clamp_exp(r, is_exp);
      }
      return nan;
    }
  }
};

}  // namespace pychain_hip
#endif
