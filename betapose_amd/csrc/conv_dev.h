// Device-side helpers shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_w64.hip): vector types,
// the element-wise epilogue for the store modes the staged 16-B path does not cover, raw-buffer loads, exact fast division.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>
#include <utility>

#include "bp_common.h"

// The timing ablations (BP_ABLATE_*: parts of a kernel compiled out, WRONG results) and the measured-and-superseded
// kernels exist only in the experimental library: `python -m betapose_amd.build --experimental` defines BP_EXPERIMENTAL.
#if !defined(BP_EXPERIMENTAL) && (defined(BP_ABLATE_SPLIT) || defined(BP_ABLATE_MFMA) || defined(BP_ABLATE_KLOOP) || defined(BP_ABLATE_PREFETCH) || \
    defined(BP_ABLATE_TAIL) || defined(BP_ABLATE_SLABSTORE) || defined(BP_ABLATE_ACKWAIT) || defined(BP_ABLATE_REDUCE) || defined(BP_ABLATE_EPILOGUE) || \
    defined(BP_W64_DEBUG))
#error "BP_ABLATE_* / BP_W64_DEBUG are timing experiments with wrong results: build them with -DBP_EXPERIMENTAL only (build.py --experimental)"
#endif

namespace bp {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}) -- for bodies whose
// inner loop bounds and register-array indices must be constants (a 48-trip `#pragma unroll` body was left rolled by
// hipcc, which put the operand fragments in scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// native vector types: they stay in VGPRs (HIP's float4 struct made hipcc park the prefetch registers in scratch)
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

static constexpr int BK = 32;
static constexpr int LDS_LD = 36;
static constexpr unsigned OOB = 0x7fffff00u;   // byte offset beyond any descriptor's num_records -> load returns 0

// ---- operand planes of an output (ConvParams::out16): what the NEXT convolution's matrix cores consume, written by
// the producer so that no consumer converts or splits anything (the reference's half path converts the activations
// too: train_YOLO/src/convolutional_kernels.cu:87,98,268-280 cuda_f32_to_f16 -> fp16 conv -> cuda_f16_to_f32).
//   np == 1: one fp16 plane (RNE);  np == 3: three bf16 planes with x == p0 + p1 + p2 exactly (8 + 8 + 8 significand bits)
struct PlaneDesc {
    __amdgpu_buffer_rsrc_t r0, r1, r2;   // one descriptor per plane, each ending at row M: rows past the tensor are dropped
    int np;
    bool f32;                            // the fp32 tensor is stored too (false: every reader takes the planes)
};
// (P = ConvParams, or the same struct in the constant address space: the persistent kernel of mega.inc reads its launch descriptors from memory)
template <class P>
__device__ __forceinline__ PlaneDesc make_plane_desc(const P& p) {
    PlaneDesc d;
    unsigned short* base = p.out16 ? p.out16 : reinterpret_cast<unsigned short*>(p.out);
    const int bytes = (int)min((long long)p.M * p.out_ld * ((p.store_mode == ST_PIXSHUF || p.store_mode == ST_UP2) ? 8 : 2), (long long)0x7fffff00);   // (a PixelShuffle output has 4 M pixels of out_ld channels)
    d.np = p.out16 ? p.out_np : 0;
    d.f32 = !(p.out16 && p.skip_f32);
    d.r0 = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
    d.r1 = __builtin_amdgcn_make_buffer_rsrc(base + (d.np == 3 ? p.out16_plane : 0), 0, bytes, 0x00020000);
    d.r2 = __builtin_amdgcn_make_buffer_rsrc(base + (d.np == 3 ? 2 * p.out16_plane : 0), 0, bytes, 0x00020000);
    return d;
}
// four consecutive channels of one pixel -> 8 B per plane at byte offset `off` (= element index * 2)
// one LDS-DMA instruction: 64 lanes x 16 B, LDS destination lane-linear from `lds`
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds, 16, (int)voff, soff, 0, 0);
}

// xcd_home launches keep their split-K hand-off inside one XCD's L2: the block must really sit on the XCD its index says
// (round-robin dispatch).  A violation would mean partial sums read from the wrong L2 -- silently wrong numbers -- so every
// launch checks it and reports through an error word (xcd_home_verify below).
__device__ __forceinline__ int xcc_id() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return (int)(id & 15);
}
// at block start: publish this slice's XCD (a write-through store nobody waits for; it is acknowledged before the block's
// ticket, which follows an s_waitcnt vmcnt(0))
template <class P>
__device__ __forceinline__ void xcd_home_mark(const P& p, int tile_id, int split) {
    // (xcd_home == 3: fault injection for the tests of the error path -- slice 0 publishes a wrong XCD)
    if (threadIdx.x == 0) __hip_atomic_store(&p.xcc_of[tile_id * 64 + split], xcc_id() ^ ((p.xcd_home >> 1) & (split == 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// in the reducing block: every slice of the tile must have run on this block's XCD.  A mismatch means the partial sums this block
// is about to read may still sit in ANOTHER XCD's L2: the block raises the engine's error word (ConvParams::err_word, agent scope) and
// stores nothing for the tile -- no trap (round 4 killed the whole HIP context, every stream of the process, on a mismatch): the host
// reads the word behind the frame (Net::take_xcd_errors; FramePipeline.run), switches the latency mode off and runs the frame again on
// the ordinary hand-off.  Returns false on a mismatch (block-uniform).
// `flag`: the block's "I am the last slice" word in LDS (holds 1 here) -- reused for the block-wide verdict, because a reduction builtin
// (__syncthreads_or) brings its own static LDS scratch into EVERY kernel that includes the tail, and the halo kernels' dynamic LDS
// opt-in of 160 KB - 64 B then exceeds the CU's 160 KB (hipFuncSetAttribute: invalid argument; found on the first GPU run of round 5)
template <class P>
__device__ __forceinline__ bool xcd_home_verify(const P& p, int tile_id, int* flag) {
    const bool bad = (int)threadIdx.x < p.splits &&
        __hip_atomic_load(&p.xcc_of[tile_id * 64 + (int)threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != xcc_id();
    if (bad) *flag = 3;                 // (every offender stores the same value)
    __syncthreads();
    const bool any_bad = *flag != 1;
    if (any_bad && threadIdx.x == 0 && p.err_word) __hip_atomic_fetch_or(p.err_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return !any_bad;
}

// A block past the work grid (ConvParams::pf_*): pull its share of the next layer's filters through the memory hierarchy.
// The bytes are dropped into 1 KB of LDS per wave (LDS-DMA: no registers, nothing for the compiler to discard); the wave
// ends when they have arrived.
template <int NT, class P>
__device__ __forceinline__ void prefetch_block(const P& p, char* lds, const int bp_bid) {
    const int e = bp_bid - p.pf_first;
    if (e < 0) return;                                   // padding between the work grid and the first prefetch block
    // pf_first is a multiple of 8: this block sits on the XCD of residue x, whose work blocks of the next launch read the
    // N-tiles n == x (mod g); it takes the ql-th (N-tile, K-slice) pair of those
    const int x = bp_bid & 7, ql = e >> 3;
    const int g = p.pf_ntn < 8 ? p.pf_ntn : 8;
    if (ql >= (p.pf_ntn / g) * p.pf_splits) return;
    const int j = ql / p.pf_splits, split = ql - j * p.pf_splits;
    const int tile_n = (x & (g - 1)) + g * j;
    const int c0 = split * p.pf_cps, c1 = min(p.pf_nchunks, c0 + p.pf_cps);
    if (c1 <= c0) return;
    const long long beg = (long long)tile_n * p.pf_tile_stride + (long long)c0 * p.pf_chunk_bytes;
    const unsigned n = (unsigned)min((long long)(c1 - c0) * p.pf_chunk_bytes, (long long)p.pf_cap);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p.pf_ptr)) + beg, 0, (int)n, 0x00020000);
    char* const dst = lds + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * 1024;
    for (unsigned off = threadIdx.x * 16u; off < n; off += NT * 16u) dma16(r, dst, off, 0);
}

// (write-through `sc1` epilogue stores -- lines leave the XCD's L2 instead of staying dirty until the kernel ends -- were
// A/B-tested in rounds 2 and 3, per layer at batch 28 and in the pipeline in every mode: no difference beyond +-0.5 %)
__device__ __forceinline__ void store_b64_wt(u32x2 v, __amdgpu_buffer_rsrc_t r, unsigned off, bool) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, 0);
}
__device__ __forceinline__ void emit_planes4(const PlaneDesc& d, f32x4 v, unsigned off) {
    if (d.np == 1) {
        const f16x4 h = __builtin_convertvector(v, f16x4);
        store_b64_wt(__builtin_bit_cast(u32x2, h), d.r0, off, false);
    } else if (d.np == 3) {
        const bf16x4 h1 = __builtin_convertvector(v, bf16x4);
        const f32x4 r1 = v - __builtin_convertvector(h1, f32x4);
        const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);
        const f32x4 r2 = r1 - __builtin_convertvector(h2, f32x4);
        const bf16x4 h3 = __builtin_convertvector(r2, bf16x4);
        store_b64_wt(__builtin_bit_cast(u32x2, h1), d.r0, off, false);
        store_b64_wt(__builtin_bit_cast(u32x2, h2), d.r1, off, false);
        store_b64_wt(__builtin_bit_cast(u32x2, h3), d.r2, off, false);
    }
}
// one element (store modes / alignments the 16-B path does not cover); idx = element index inside the output view
template <class P>
__device__ __forceinline__ void emit_plane1(const P& p, long long idx, float v) {
    if (!p.out16) return;
    if (p.out_np == 1) {
        const _Float16 h = (_Float16)v;
        p.out16[idx] = __builtin_bit_cast(unsigned short, h);
    } else if (p.out_np == 3) {
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        const __bf16 h3 = (__bf16)r2;
        p.out16[idx] = __builtin_bit_cast(unsigned short, h1);
        p.out16[idx + p.out16_plane] = __builtin_bit_cast(unsigned short, h2);
        p.out16[idx + 2 * p.out16_plane] = __builtin_bit_cast(unsigned short, h3);
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// v = raw accumulator for output element (m, n); bias = p.bias[n] (loaded once per lane by the caller: the
// epilogue's stores may alias p.bias as far as the compiler knows, so an in-loop load is re-issued and waited
// for after every store -- 16 serialized L2 round trips per tile, measured as the dominant cost of short layers)
template <class P>
__device__ __forceinline__ void epilogue_store(const P& p, int m, int n, float v, float bias) {
    v += bias;
    int b = 0, pix = m;
    const int hw = p.OH * p.OW;
    const bool need_pix = p.store_mode != ST_NHWC || p.res_scale != nullptr;
    if (need_pix) {
        b = m / hw;
        pix = m - b * hw;
    }
    float r = 0.f;
    if (p.res) {
        if (p.res16) { const _Float16 h = __builtin_bit_cast(_Float16, p.res16[(long long)m * p.res_ld + n]); r = (float)h; }
        else r = p.res[(long long)m * p.res_ld + n];
        if (p.res_scale) r *= p.res_scale[b * p.Cout + n];
    }
    if (!p.res_after_act) v += r;
    v = apply_act(v, p.act);
    if (p.res_after_act) v += r;
    const bool keep32 = !(p.out16 && p.skip_f32);
    switch (p.store_mode) {
        case ST_NHWC:
            if (keep32) p.out[(long long)m * p.out_ld + n] = v;
            emit_plane1(p, (long long)m * p.out_ld + n, v);
            break;
        case ST_UP2: {
            const int oy = pix / p.OW, ox = pix - oy * p.OW;
            const int W2 = 2 * p.OW;
            const long long i0 = ((long long)(b * 2 * p.OH + 2 * oy) * W2 + 2 * ox) * p.out_ld + n;
            float* o = p.out + i0;
            if (keep32) {
                o[0] = v;
                o[p.out_ld] = v;
                o[(long long)W2 * p.out_ld] = v;
                o[(long long)(W2 + 1) * p.out_ld] = v;
            }
            emit_plane1(p, i0, v);
            emit_plane1(p, i0 + p.out_ld, v);
            emit_plane1(p, i0 + (long long)W2 * p.out_ld, v);
            emit_plane1(p, i0 + (long long)(W2 + 1) * p.out_ld, v);
        } break;
        case ST_PIXSHUF: {
            const int oy = pix / p.OW, ox = pix - oy * p.OW;
            const int cq = p.Cout >> 2;
            const int ij = n / cq, c = n - ij * cq;
            const int y = 2 * oy + (ij >> 1), x = 2 * ox + (ij & 1);
            const long long i0 = ((long long)(b * 2 * p.OH + y) * (2 * p.OW) + x) * p.out_ld + c;
            if (keep32) p.out[i0] = v;
            emit_plane1(p, i0, v);
        } break;
        case ST_NCHW:
            p.out[((long long)b * p.Cout + n) * hw + pix] = v;
            break;
    }
}

// 16 accumulators of one lane: column n, rows m_base + (r&3) + 8*(r>>2) -- element-wise epilogue for the store
// modes / alignments the staged float4 path does not cover (heads with 18 channels, upsample, PixelShuffle, NCHW)
template <class P>
__device__ __forceinline__ void epilogue_tile(const P& p, const f32x16& v, int m_base, int n) {
    if (n >= p.Cout) return;
    const float bias = p.bias[n];
    static_for<16>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        const int m = m_base + (e & 3) + 8 * (e >> 2);
        if (m < p.M) epilogue_store(p, m, n, v[e], bias);
    });
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, 0);
    return f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
}

// four consecutive residual channels: 16 B of the fp32 tensor, or 8 B of its fp16 plane widened (ConvParams::res16); `off` = byte offset
// in the respective buffer
__device__ __forceinline__ f32x4 load_res4(__amdgpu_buffer_rsrc_t rsrc, unsigned off, bool r16) {
    if (r16) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
        return __builtin_convertvector(__builtin_bit_cast(f16x4, t), f32x4);
    }
    return buf_load4(rsrc, off, 0);
}

// a - b as ONE v_sub_f32.  Opaque to the optimizer on purpose: hipcc packs neighbouring f32 subtractions into
// v_pk_add_f32, which is slow beside MFMAs (MI355X_MICROARCH.md, price of fillers).  Kept in a __device__ function: an
// asm statement with register constraints directly in a __global__ template silently drops the kernel's host stub.
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// exact for 0 <= m < 2^24, d > 0: quotient by float reciprocal + one correction step (a 32-bit integer division
// costs ~40 VALU instructions; the tile prologue needs 2 per row)
__device__ __forceinline__ int fast_div(int m, int d, float rcp) {
    int q = (int)((float)m * rcp);
    const int r = m - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}


// PixelShuffle(2) stores through the staged epilogue: tile row -> byte offset of the thread's four channels in the [2 OH][2 OW][Cout / 4]
// output (filters pre-permuted: column n' = (i 2 + j) (Cout / 4) + c).  Computed per pass (two exact divisions) rather than kept as a
// table: four more live registers cost the 64x64 filters-direct kernel its fourth wave per SIMD (132 against 128 VGPRs).
struct PixShufRows {
    int m_first, m_step, M, hw, OW, OH, out_ld;
    unsigned sub;             // ((i 2 OW + j) out_ld + c) 4: the sub-pixel and channel part
    float rcp_hw, rcp_ow;
    __device__ __forceinline__ unsigned offset(int pass) const {
        const int m = m_first + pass * m_step;
        if (m >= M) return OOB;
        const int q = (int)((float)m * rcp_hw);
        int b = q; { const int r = m - q * hw; if (r < 0) --b; else if (r >= hw) ++b; }
        const int rem = m - b * hw;
        int oy = (int)((float)rem * rcp_ow); { const int r = rem - oy * OW; if (r < 0) --oy; else if (r >= OW) ++oy; }
        const int ox = rem - oy * OW;
        return (unsigned)((((b * 2 * OH + 2 * oy) * (2 * OW) + 2 * ox) * out_ld) * 4) + sub;
    }
};

// Nearest x2 upsample stores (ST_UP2, the two route branches of YOLOv3: yolo/darknet.py:273-276 nn.Upsample behind a 1x1 convolution) through
// the staged epilogue: tile row -> byte offset of the thread's four channels at output pixel (2 oy, 2 ox); the other three copies sit one pixel
// right / one row down.  (Round 5: on the element-wise path these two layers carried a 6-13 us tail at batch 1 and ran 38 us against 17 us
// for their sister layers at batch 28.)
struct Up2Rows {
    int m_first, m_step, M, hw, OW, OH, out_ld;
    unsigned sub;             // channel part, bytes
    float rcp_hw, rcp_ow;
    __device__ __forceinline__ unsigned offset(int pass) const {
        const int m = m_first + pass * m_step;
        if (m >= M) return OOB;
        const int b = fast_div(m, hw, rcp_hw);
        const int rem = m - b * hw;
        const int oy = fast_div(rem, OW, rcp_ow);
        const int ox = rem - oy * OW;
        return (unsigned)((((b * 2 * OH + 2 * oy) * (2 * OW) + 2 * ox) * out_ld) * 4) + sub;
    }
};

// Row loop of the staged (16-B per lane) epilogue: each pass reads 4 consecutive channels of one tile row from the
// LDS staging tile, applies bias / residual / activation and stores 16 B.  ACT and RES are compile-time (the caller
// switches once per block): RES 0 none, 1 add before the activation (ResNet), 2 add after it (YOLO shortcut).
// Output and residual go through raw buffer descriptors whose range ends at row M, so rows past the tensor are dropped
// by the hardware -- no per-pass branch.
// SE channel scale of the residual (ConvParams::res_scale [N][Cout], the SE blocks' "downsample" convolutions: out = act(conv + T * y),
// SE_Resnet.py:31-40 with SELayer): four channels of the row's IMAGE -- rows of a tile may belong to different images of a batch.  Until
// round 5 these four layers took the element-wise epilogue: 168 / 92 / 60 / 73 us at batch 28 (fp16) where their sister 1x1 layers take
// 63 / 35 / 23 / 17, and 18.5 us against 8.7 at batch 1.
struct ResScaleRows {
    const float* rs;          // res_scale + the thread's first channel
    int m_first, m_step, hw, Cout, m_last;
    float rcp_hw;
    // (A form that requested the first and last pass's image scales once, ahead of the row loop, and selected per pass was SLOWER than this
    // load per pass -- 120.8 / 73.1 / 43.7 / 38.7 us against 106.1 / 62.8 / 41.3 / 35.9 us for the four layers at batch 28: the loads hit the
    // L1 and the selects serialised the loop; removed.)
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ f32x4 at(int pass) const {
        const int m = min(m_first + pass * m_step, m_last);      // (rows past M read the last image's scale; their store is dropped)
        return *reinterpret_cast<const f32x4*>(rs + fast_div(m, hw, rcp_hw) * Cout);
    }
};

template <int ACT, int RES, int PASSES, int PF>
__device__ __forceinline__ void epilogue_rows(const float* srow, int s_step, f32x4 bias4,
                                              __amdgpu_buffer_rsrc_t rsrcO, unsigned off_o, unsigned step_o,
                                              __amdgpu_buffer_rsrc_t rsrcR, unsigned off_r, unsigned step_r, const f32x4* rpre,
                                              const PlaneDesc& pd, bool r16 = false, const PixShufRows* ps = nullptr, const ResScaleRows* rsc = nullptr,
                                              const Up2Rows* up = nullptr) {
    // ps (PixelShuffle stores, conv_tail.inc): the output byte offset of a pass comes from its row's pixel instead of off_o + pass * step_o
    // PF: the residual rows were requested before the tile was staged (rpre[pass], registers): a cold 16-B load costs the
    // block > 1 us at the very end of the kernel otherwise
    // Big tiles (8 or more passes: the 128-row plane tiles of the batched runs) that could not hold their residual rows in registers through
    // the K loop request them here in GROUPS of eight passes ahead of the group's stores: with one request per pass, issued where it is used,
    // every pass paid a full memory round trip behind the previous pass's store (the counter is in order: the load waits for the store's
    // acknowledgement too) -- 13 us of the 71 us of a 52x52 128 -> 256 layer with a skip connection at batch 28 (round 5).  The accumulators
    // are staged by now, so the registers are free.
    constexpr int GRP = (RES != 0 && !PF && PASSES >= 8) ? 8 : 1;
    u32x4 rgrp[GRP];
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if constexpr (GRP > 1) {
            if (pass % GRP == 0) {
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    if (pass + g < PASSES) {
                        const unsigned o = off_r + (unsigned)g * step_r;
                        if (r16) { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)o, 0, 0); rgrp[g] = u32x4{t.x, t.y, 0u, 0u}; }
                        else rgrp[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)o, 0, 0);
                    }
                }
            }
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(srow);
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES != 0) {
            if constexpr (PF) r4 = rpre[pass];
            else if constexpr (GRP > 1) {
                const u32x4 t = rgrp[pass % GRP];
                if (r16) r4 = __builtin_convertvector(__builtin_bit_cast(f16x4, u32x2{t.x, t.y}), f32x4);
                else r4 = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
            } else r4 = load_res4(rsrcR, off_r, r16);
            if (rsc) r4 *= rsc->at(pass);
        }
        v += bias4;
        if constexpr (RES == 1) v += r4;
        if constexpr (ACT == ACT_LEAKY) {
            v.x = v.x > 0.f ? v.x : 0.1f * v.x; v.y = v.y > 0.f ? v.y : 0.1f * v.y;
            v.z = v.z > 0.f ? v.z : 0.1f * v.z; v.w = v.w > 0.f ? v.w : 0.1f * v.w;
        } else if constexpr (ACT == ACT_RELU) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
            v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        }
        if constexpr (RES == 2) v += r4;
        const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        if (ps) off_o = ps->offset(pass);
        if (up) {
            off_o = up->offset(pass);
            const unsigned right = (unsigned)(up->out_ld * 4), down = (unsigned)(2 * up->OW * up->out_ld * 4);
            if (off_o != OOB) {      // (OOB + an increment could wrap into range)
                if (pd.f32) {
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)(off_o + right), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)(off_o + down), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)(off_o + down + right), 0, 0);
                }
                emit_planes4(pd, v, (off_o + right) >> 1);
                emit_planes4(pd, v, (off_o + down) >> 1);
                emit_planes4(pd, v, (off_o + down + right) >> 1);
            }
        }
        if (pd.f32) __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)off_o, 0, 0);
        emit_planes4(pd, v, off_o >> 1);      // the planes mirror the fp32 view: same element index, half the bytes
        srow += s_step;
        off_o += step_o;
        off_r += step_r;
    }
}

// LDS rows of the 16-bit kernels are 32 elements = 64 B, unpadded; the four 16-B granules of row r are stored at
// granule index g ^ ((r >> 1) & 3), so the 16-B fragment reads of 8 consecutive lanes (8 rows, same logical granule)
// hit all 32 banks once
static constexpr int LDH = 32;

template <int NP> struct HalfOps;
template <> struct HalfOps<1> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct HalfOps<3> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

}  // namespace bp
