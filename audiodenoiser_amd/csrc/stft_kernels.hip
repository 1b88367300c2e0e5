// STFT magnitude for gfx950: framing + periodic-Hann window + real FFT + |.| fused in one kernel.
//
// Replaces librosa.stft + librosa.magphase as driven by the reference's
//   audio_to_magnitude_spectrogram  (/root/reference/code/create_train_dataset.py:162-174, center=False)
//   audio_to_spectrogram            (/root/reference/code/create_test_dataset.py:35-41,    center=True)
// Semantics (librosa 0.10): window = periodic Hann of length n_fft, centre padding = n_fft/2 ZEROS each side,
// n_frames = 1 + (L_padded - n_fft) / hop, output (n_fft/2+1, n_frames) with the frame index fastest.
//
// Algorithm: the n_fft real samples of a frame are packed as M = n_fft/2 complex points z[n] = x[2n] + i x[2n+1];
// an M-point Stockham autosort FFT runs with 8 points per thread in registers (radix-8 passes, one radix-4/2
// tail pass), exchanging through LDS between passes; the real spectrum follows from
//   X[k]   = Ev[k] + w^k Od[k],  X[M-k] = conj(Ev[k] - w^k Od[k]),  Ev = (Z[k]+conj(Z[M-k]))/2,
//   Od = (Z[k]-conj(Z[M-k]))/(2i),  w = exp(-2 pi i / n_fft).
// HBM-bound by design (about 10 FLOP/B): a workgroup owns FPB consecutive frames of one clip, stages the
// overlapping audio span in LDS once (75 % overlap at hop = n_fft/4 is served on-chip), and transposes the
// magnitudes through LDS so that stores run along the frame axis.
#include "adn_internal.h"
#include "fft_core.h"

#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace adn {
namespace {

using namespace fftcore;

// Where and how the magnitudes are stored.  Plain STFT: rows = n_fft/2+1, row_stride = n_frames, clip_stride = rows *
// n_frames, quantize = 0.  Fused loader rule (adn_stft_mag_fit): the (H, W) window of SpectrogramDataset's
// _pad_or_truncate (data_loader.py:54-72) -- only rows < H and frames < W are computed / stored, at row stride W --
// and every value goes through the fp16 round trip of data_loader.py:41-42.
struct StftOut {
    long row_stride, clip_stride;
    int rows;
    int quantize;
};

__device__ __forceinline__ float stft_emit(float v, int quantize)
{
    return quantize ? __half2float(__float2half_rn(v)) : v;
}

template <int M>
struct StftCfg {
    static constexpr int N = 2 * M;
    static constexpr int TPF = M / 8;                                  // threads per frame
    static constexpr int FPW = STFT_THREADS / TPF;                      // frames the workgroup can run at once
    static constexpr int FPB = (M <= 512) ? (FPW > 16 ? FPW : 16) : 8192 / M;   // frames per workgroup
    static constexpr int FB = FPW < FPB ? FPW : FPB;                    // frames per batch
    static constexpr int NBATCH = FPB / FB;
    static constexpr int MAGSTR = FPB + 1;
};

// tables (device, fp32, computed in double on the host): win[N], tw[M] = exp(-2 pi i j / M),
// tw2[M/2+1] = exp(-2 pi i k / N)
template <int M>
__global__ __launch_bounds__(STFT_THREADS) void stft_mag_kernel(const float *__restrict__ audio, long L, int hop, int pad,
                                                               long n_frames, int groups_per_clip,
                                                               const float *__restrict__ tables, float *__restrict__ out,
                                                               int work_floats, const StftOut o)
{
    using C = StftCfg<M>;
    constexpr int N = C::N, TPF = C::TPF, FPB = C::FPB, FB = C::FB, NBATCH = C::NBATCH, MAGSTR = C::MAGSTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem;                                             // N
    float2 *s_tw = reinterpret_cast<float2 *>(smem + N);             // M
    float2 *s_tw2 = s_tw + M;                                        // M/2 + 1 (padded to even count)
    float *s_mag = smem + N + 2 * M + (M + 2);                       // (M+1) * MAGSTR
    float *s_work = s_mag + (M + 1) * MAGSTR + (((M + 1) * MAGSTR) & 1);   // keep 8-byte alignment
    float2 *s_sc = reinterpret_cast<float2 *>(s_work);

    const int tid = threadIdx.x;
    const long clip = blockIdx.x / groups_per_clip;
    const int grp = blockIdx.x - (int)(clip * groups_per_clip);
    const long f0 = (long)grp * FPB;
    const float *aud = audio + clip * L;

    for (int i = tid; i < N + 2 * M + (M + 2); i += STFT_THREADS) smem[i] = tables[i];

    const int fl = tid / TPF, t = tid - fl * TPF;
    const int span = (FB - 1) * hop + N;
    (void)work_floats;

#pragma unroll 1
    for (int bt = 0; bt < NBATCH; ++bt) {
        const long fb0 = f0 + (long)bt * FB;
        // ---- stage the audio span of this batch (zero outside the clip: centre padding / tail) ----
        const long s0 = fb0 * hop - pad;
        for (int i = tid; i < span; i += STFT_THREADS) {
            const long s = s0 + i;
            s_work[i] = (s >= 0 && s < L) ? aud[s] : 0.f;
        }
        __syncthreads();
        float2 v[8];
        if (fl < FB) {
            const float *a = s_work + fl * hop;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n2 = 2 * (t + u * TPF);
                v[u] = make_float2(s_win[n2] * a[n2], s_win[n2 + 1] * a[n2 + 1]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = make_float2(0.f, 0.f);
        }
        // frames beyond FB (possible only when FPW > FPB, never with the configs above) share slot 0 harmlessly
        float2 *sc = s_sc + (fl < FB ? fl : 0) * M;
        fft_frame<M>(sc, s_tw, t, v);      // first pass syncs before it overwrites the audio span

        // ---- real-FFT post-processing + magnitude, into the [bin][frame] LDS image ----
        if (fl < FB) {
            const int fcol = bt * FB + fl;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = t + b * TPF;                 // 0 .. M/2-1
                if (k == 0) {
                    const float2 z0 = sc[0];
                    s_mag[0 * MAGSTR + fcol] = fabsf(z0.x + z0.y);
                    s_mag[M * MAGSTR + fcol] = fabsf(z0.x - z0.y);
                    const float2 zh = sc[M / 2];
                    s_mag[(M / 2) * MAGSTR + fcol] = sqrtf(zh.x * zh.x + zh.y * zh.y);
                } else {
                    const float2 A = sc[k];
                    const float2 Bc = sc[M - k];
                    const float2 Bz = make_float2(Bc.x, -Bc.y);
                    const float2 ev = make_float2(0.5f * (A.x + Bz.x), 0.5f * (A.y + Bz.y));
                    const float2 d = make_float2(0.5f * (A.x - Bz.x), 0.5f * (A.y - Bz.y));
                    const float2 od = make_float2(d.y, -d.x);          // d / i
                    const float2 wo = cmul(s_tw2[k], od);
                    const float2 xa = cadd(ev, wo), xb = csub(ev, wo);
                    s_mag[k * MAGSTR + fcol] = sqrtf(xa.x * xa.x + xa.y * xa.y);
                    s_mag[(M - k) * MAGSTR + fcol] = sqrtf(xb.x * xb.x + xb.y * xb.y);
                }
            }
        }
        __syncthreads();   // scratch is restaged by the next batch; s_mag complete after the last one
    }

    // ---- store: lanes run along the frame axis ----
    float *ob = out + clip * o.clip_stride;
    for (int idx = tid; idx < (M + 1) * FPB; idx += STFT_THREADS) {
        const int k = idx / FPB, f = idx - k * FPB;
        if (f0 + f < n_frames && k < o.rows) ob[(long)k * o.row_stride + f0 + f] = stft_emit(s_mag[k * MAGSTR + f], o.quantize);
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-synchronous variant for n_fft <= 1024 (a frame fits one wave: M/8 <= 64 threads).
//
// No workgroup barrier inside the FFT: a frame's 8-point-per-thread exchanges go through a private LDS
// slot and rely on the in-order execution of one wave's DS instructions, so waves drift freely and hide
// each other's global/LDS latency.  All per-thread constants (window, twiddles of every pass, the real-FFT
// post-twiddles) live in registers, loaded once per workgroup.  Frames are read straight from global memory
// (8-byte coalesced loads; the 75 % overlap between consecutive frames is served by L1/L2), the exchange image
// is padded by one element per 8 (conflict-free ds_write_b64 / ds_read_b64), and magnitudes are transposed
// through an LDS [bin][frame] image so that HBM stores run along the frame axis.
// ------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// volatile LDS views: keep every exchange access a single ds_read_b64 / ds_write_b64 -- the merged two-address
// forms (ds_read2_b64, ds_read2st64_b64) are served 16 lanes at a time over 32 banks at half the rate and would
// break the bank analysis of xidx()
typedef volatile v2f __attribute__((address_space(3))) lds_v_v2f;
__device__ __forceinline__ v2f ADN_XRD(const v2f *p) { return *(const lds_v_v2f *)p; }
__device__ __forceinline__ void ADN_XWR(v2f *p, v2f val) { *(lds_v_v2f *)p = val; }

// Complex arithmetic on the packed-fp32 pipe with the rotations folded into the VOP3P operand modifiers (op_sel picks which
// half of a 64-bit source feeds the low result, op_sel_hi the high result; neg_lo / neg_hi negate the selected half).  hipcc
// does not form these from vector code: a multiplication by -i or a conjugate became v_mov / v_xor pairs in front of a plain
// v_pk_add_f32 -- 85 v_mov per frame against 154 packed operations.  (ADN_STFT_OPSEL=0: the plain vector-code form.)
#ifndef ADN_STFT_OPSEL
#define ADN_STFT_OPSEL 1
#endif
#if !ADN_STFT_OPSEL
__device__ __forceinline__ v2f vnegi(v2f a) { return v2f{a.y, -a.x}; }
#endif
__device__ __forceinline__ v2f vadd_negi(v2f a, v2f b)   // a + (-i) b = (a.x + b.y, a.y - b.x)
{
#if ADN_STFT_OPSEL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + vnegi(b);
#endif
}
__device__ __forceinline__ v2f vsub_negi(v2f a, v2f b)   // a - (-i) b = (a.x - b.y, a.y + b.x)
{
#if ADN_STFT_OPSEL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - vnegi(b);
#endif
}
__device__ __forceinline__ v2f vadd_conj(v2f a, v2f b)   // a + conj(b)
{
#if ADN_STFT_OPSEL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + v2f{b.x, -b.y};
#endif
}
__device__ __forceinline__ v2f vsub_conj(v2f a, v2f b)   // a - conj(b)
{
#if ADN_STFT_OPSEL
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - v2f{b.x, -b.y};
#endif
}
__device__ __forceinline__ v2f vmul(v2f a, v2f b)   // complex multiply a * b = a.x * (b.x, b.y) + a.y * (-b.y, b.x)
{
#if ADN_STFT_OPSEL
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
#else
    const v2f bs = {-b.y, b.x};
    return a.x * b + a.y * bs;
#endif
}
__device__ __forceinline__ v2f vmul_negi(v2f a, v2f d)   // a * ((-i) d) = a.x * (d.y, -d.x) + a.y * (d.x, d.y)
{
#if ADN_STFT_OPSEL
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(d));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(d), "v"(t));
    return r;
#else
    const v2f b = vnegi(d);
    const v2f bs = {-b.y, b.x};
    return a.x * b + a.y * bs;
#endif
}

template <int R>
__device__ __forceinline__ void vdft(v2f *v);
template <>
__device__ __forceinline__ void vdft<2>(v2f *v)
{
    const v2f a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
}
template <>
__device__ __forceinline__ void vdft<4>(v2f *v)
{
    const v2f t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], d13 = v[1] - v[3];
    v[0] = t0 + t2;
    v[1] = vadd_negi(t1, d13);
    v[2] = t0 - t2;
    v[3] = vsub_negi(t1, d13);
}
template <>
__device__ __forceinline__ void vdft<8>(v2f *v)
{
    v2f e[4] = {v[0], v[2], v[4], v[6]};
    v2f o[4] = {v[1], v[3], v[5], v[7]};
    vdft<4>(e);
    vdft<4>(o);
    const float s = 0.70710678118654752440f;
    // o1 * w8 = s (o1.x + o1.y, o1.y - o1.x);  o2 * w8^2 = -i o2;  o3 * w8^3 = -i * s (o3.x + o3.y, o3.y - o3.x)
    const v2f u1 = s * vadd_negi(o[1], o[1]), u3 = s * vadd_negi(o[3], o[3]);
    v[0] = e[0] + o[0];
    v[4] = e[0] - o[0];
    v[1] = e[1] + u1;
    v[5] = e[1] - u1;
    v[2] = vadd_negi(e[2], o[2]);
    v[6] = vsub_negi(e[2], o[2]);
    v[3] = vadd_negi(e[3], u3);
    v[7] = vsub_negi(e[3], u3);
}

// Exchange-image layouts.  X = 1: written by pass 1, read by pass 2; X = 2: pass 2 -> pass 3; X = 3: last pass ->
// real-FFT post-processing.  A ds_read_b64 is served in two groups of 32 lanes over 64 dword banks, a ds_write_b64 in
// four groups of 16 lanes over 32 banks (MI355X_MICROARCH.md, LDS).  Every read here is "32 lanes, 32 consecutive
// elements", so ANY permutation inside aligned 32-element blocks keeps the reads conflict-free, and the permutation
// is chosen per exchange so that the 16 lanes of a write group hit 16 distinct element residues mod 16:
//   M >= 128 (n_fft 256, 512 [the reference's], 1024 [BASELINE's]; 32 or 64 lanes per frame):
//     X = 1: writes 8t+q          -> e'[1:0] = e[4:3], e'[3:2] = e[1:0] ^ e[6:5], e'[4] = e[2]
//     X = 2: writes 64(t>>3)+(t&7)+8q -> e' = e ^ (e[6] << 3)
//     X = 3: writes t+64q, reads k and M-k -> identity
//   smaller sizes: one pad element per 8 (reads of the later passes are then 2-way conflicted).
template <int M, int X>
__device__ __forceinline__ int xidx(int e)
{
    if constexpr (M >= 128) {
        if constexpr (X == 1) return (e & ~31) | ((e & 4) << 2) | ((((e & 3) ^ ((e >> 5) & 3))) << 2) | ((e >> 3) & 3);
        else if constexpr (X == 2) return e ^ ((e >> 3) & 8);
        else return e;
    } else {
        return e + (e >> 3);
    }
}
template <int M> constexpr int exch_size() { return M >= 128 ? M : M + M / 8; }

__device__ __forceinline__ void wave_lds_fence()
{
    // order this wave's LDS accesses for the compiler; the hardware runs one wave's DS ops in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One Stockham pass (radix R, P = product of earlier radices) with the twiddles in registers `tw`
// ((R-1) per butterfly, butterfly-major).
template <int M, int R, int P, int XIN, int XOUT>
__device__ __forceinline__ void wave_pass(v2f *sc, const v2f *tw, int t, v2f *v)
{
    constexpr int TPF = M / 8, NBF = 8 / R, T = M / R;
#pragma unroll
    for (int b = 0; b < NBF; ++b)
#pragma unroll
        for (int u = 0; u < R; ++u) v[b * R + u] = ADN_XRD(sc + xidx<M, XIN>(t + b * TPF + u * T));
    wave_lds_fence();
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int i = t + b * TPF;
        const int k = i & (P - 1);
#pragma unroll
        for (int u = 1; u < R; ++u) v[b * R + u] = vmul(v[b * R + u], tw[b * (R - 1) + u - 1]);
        vdft<R>(v + b * R);
        const int j = (i - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) ADN_XWR(sc + xidx<M, XOUT>(j + q * P), v[b * R + q]);
    }
    wave_lds_fence();
}

template <int M, int R, int P>
__device__ __forceinline__ void load_pass_twiddles(const float *tables, int t, v2f *tw)
{
    constexpr int TPF = M / 8, NBF = 8 / R, N = 2 * M;
    const v2f *gtw = reinterpret_cast<const v2f *>(tables + N);
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int k = (t + b * TPF) & (P - 1);
#pragma unroll
        for (int u = 1; u < R; ++u) tw[b * (R - 1) + u - 1] = gtw[(u * k * (M / (P * R))) & (M - 1)];
    }
}

// radix plan: pass 1 is always radix 8 (P=1, no twiddles); then (R2,P2=8) and (R3,P3=8*R2) where present
template <int M> struct WavePlan;
template <> struct WavePlan<32>  { static constexpr int R2 = 4, R3 = 1; };
template <> struct WavePlan<64>  { static constexpr int R2 = 8, R3 = 1; };
template <> struct WavePlan<128> { static constexpr int R2 = 8, R3 = 2; };
template <> struct WavePlan<256> { static constexpr int R2 = 8, R3 = 4; };
template <> struct WavePlan<512> { static constexpr int R2 = 8, R3 = 8; };

// Bijective remap of the workgroup id so that workgroups placed on one XCD (ids congruent mod 8 under the
// observed round-robin placement) own neighbouring frame groups: the two halves of an output cache line and the
// overlapping audio lines then meet in ONE L2 instead of being written back / fetched partially by several.
// Affects speed only.
__device__ __forceinline__ int stft_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// at least 3 waves per SIMD: the register allocation granule is 8, so 169 VGPRs would already drop to 2
template <int M, int NW, int FPB, int WPE, bool FIT>
__global__ __launch_bounds__(NW * 64, WPE) void stft_wave_kernel(const float *__restrict__ audio, long L, int hop, int pad,
                                                             long n_frames, int groups_per_clip, int gpb,
                                                             int blocks_per_clip, const float *__restrict__ tables,
                                                             float *__restrict__ out, const StftOut o)
{
    constexpr int N = 2 * M, TPF = M / 8, NT = NW * 64;
    constexpr int SLOTS = NT / TPF, FPS = FPB / SLOTS;      // frames per slot and group, processed in sequence
    constexpr int MAGSTR = FPB + 1;
    constexpr int SCSZ = exch_size<M>();                     // exchange image (v2f elements), see xidx
    constexpr int R2 = WavePlan<M>::R2, R3 = WavePlan<M>::R3;
    static_assert(TPF <= 64 && FPB % SLOTS == 0 && FPS >= 1 && NT % FPB == 0, "bad STFT tiling");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_mag = smem;                                                   // (M+1) * MAGSTR
    v2f *s_sc = reinterpret_cast<v2f *>(smem + (((M + 1) * MAGSTR + 1) & ~1));

    const int tid = threadIdx.x;
    // a frame per wave (TPF == 64): the slot is wave-uniform -> frame indices, bounds tests and the prefetch
    // branches run on the scalar unit
    const int slot = (TPF == 64) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / TPF;
    const int t = tid - slot * TPF;
    const int lid = stft_xcd_remap(blockIdx.x, gridDim.x);
    const int clip = lid / blocks_per_clip;
    const int g_first = (lid - clip * blocks_per_clip) * gpb;
    const int g_end = min(g_first + gpb, groups_per_clip);
    const float *aud = audio + (long)clip * L;
    // FIT = false (plain STFT): rows = M + 1, row stride = n_frames, no rounding -- kept a compile-time case: the extra
    // row test and the fp16 round trip per stored value cost 9 % of the kernel when left to run-time flags
    const long row_stride = FIT ? o.row_stride : n_frames;
    const int out_rows = FIT ? o.rows : M + 1;
    float *oclip = out + (long)clip * (FIT ? o.clip_stride : (long)(M + 1) * n_frames);
    v2f *sc = s_sc + slot * SCSZ;

    const int Li = (int)L;                                // adn_stft_mag guarantees L < 2^30
    const bool base_aligned = (reinterpret_cast<uintptr_t>(aud) & 7) == 0;

    // raw (unwindowed) samples of frame `fidx` of this clip -> dst[8]
    auto load_frame = [&](int fidx, v2f *dst) {
        const int fstart = fidx * hop - pad;               // may be < 0 (centre padding) or run past the clip
        const float *ap = aud + fstart + 2 * t;
        if (fstart >= 0 && fstart + N <= Li && base_aligned && !(fstart & 1)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[u] = *reinterpret_cast<const v2f *>(ap + 2 * u * TPF);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = fstart + 2 * (t + u * TPF);
                dst[u].x = (s >= 0 && s < Li) ? ap[2 * u * TPF] : 0.f;
                dst[u].y = (s + 1 >= 0 && s + 1 < Li) ? ap[2 * u * TPF + 1] : 0.f;
            }
        }
    };

    // this slot's frame sequence: groups g_first..g_end-1, FPS consecutive frames in each
    const int n_seq = (g_end - g_first) * FPS;
    // The first frame's samples are requested BEFORE the constant set-up below: a workgroup lives for only ~5 us and
    // nothing can be prefetched for its first frame, so this HBM round trip (~2 us under load) is overlapped with the
    // table copy, its two barriers and the 26 per-lane constant reads instead of following them.
    v2f nx[8];
    load_frame(g_first * FPB + slot * FPS, nx);

    // ---- per-thread constants, kept in registers over all the workgroup's frames.  The 10 KB table is copied
    // into LDS cooperatively (the [bin][frame] image is still unused) and each lane picks its 26 values from
    // there: 10 KB of L2->CU traffic per workgroup instead of 53 KB of per-lane gathers (the kernel is bound by
    // the CU's ingest path, not by HBM).
    {
        constexpr int TBL = N + 2 * M + (M + 2);
        static_assert(TBL <= (M + 1) * MAGSTR, "table must fit the magnitude image");
        for (int i = tid * 4; i < TBL; i += NT * 4) {
            if (i + 4 <= TBL) *reinterpret_cast<f4 *>(s_mag + i) = *reinterpret_cast<const f4 *>(tables + i);
            else
                for (int j = i; j < TBL; ++j) s_mag[j] = tables[j];
        }
        __syncthreads();
    }
    v2f win[8], tw2[(R2 - 1) * (8 / R2)], tw3[R3 > 1 ? (R3 - 1) * (8 / R3) : 1], twp[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) win[u] = 0.5f * *reinterpret_cast<const v2f *>(s_mag + 2 * (t + u * TPF));   // Z/2, exact
    load_pass_twiddles<M, R2, 8>(s_mag, t, tw2);
    if constexpr (R3 > 1) load_pass_twiddles<M, R3, 8 * R2>(s_mag, t, tw3);
    {
        const v2f *g2 = reinterpret_cast<const v2f *>(s_mag + N + 2 * M);
#pragma unroll
        for (int b = 0; b < 4; ++b) twp[b] = g2[t + b * TPF];
    }
    __syncthreads();                                      // everyone holds its constants: the image may be written

    int g = g_first, fi = 0;
    // one frame of this slot's sequence (the body of the frame loop; see the two loops below)
    auto frame = [&](const int it) __attribute__((always_inline)) {
        const int fcol = slot * FPS + fi;
        v2f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = nx[u] * win[u];
        // software prefetch: the next frame's loads fly under this frame's FFT
        int gn = g, fn = fi + 1;
        if (fn == FPS) { fn = 0; ++gn; }
        if (it + 1 < n_seq) {
            const int fnx = gn * FPB + slot * FPS + fn;
            const int fs = fnx * hop - pad;
            // hop = n_fft/4 (the reference's and BASELINE's setting): the next frame of this slot starts 2*TPF
            // complex points later, i.e. its element u is this frame's element u+2 -> shift six registers and load
            // two (each audio sample is read once instead of four times)
            if (hop * 4 == N && fn != 0 && fs >= 0 && fs + N <= Li && base_aligned && !(fs & 1)) {
#pragma unroll
                for (int u = 0; u < 6; ++u) nx[u] = nx[u + 2];
                const float *ap = aud + fs + 2 * t;
                nx[6] = *reinterpret_cast<const v2f *>(ap + 2 * 6 * TPF);
                nx[7] = *reinterpret_cast<const v2f *>(ap + 2 * 7 * TPF);
            } else {
                load_frame(fnx, nx);
            }
        }

        // pass 1: radix 8, P = 1
        vdft<8>(v);
#pragma unroll
        for (int q = 0; q < 8; ++q) ADN_XWR(sc + xidx<M, 1>(8 * t + q), v[q]);
        wave_lds_fence();
        wave_pass<M, R2, 8, 1, (R3 > 1 ? 2 : 3)>(sc, tw2, t, v);
        if constexpr (R3 > 1) wave_pass<M, R3, 8 * R2, 2, 3>(sc, tw3, t, v);

        // ---- real-FFT post-processing + magnitude into the [bin][frame] image ----
        float *mg = s_mag + fcol;
        v2f pa[4], pb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {          // all reads first: the volatile views keep program order
            const int k = t + b * TPF;
            pa[b] = ADN_XRD(sc + xidx<M, 3>(k));
            pb[b] = ADN_XRD(sc + xidx<M, 3>((M - k) & (M - 1)));   // k = 0 pairs with itself
        }
        const v2f zh = 2.0f * ADN_XRD(sc + xidx<M, 3>(M / 2));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = t + b * TPF;
            const v2f A = pa[b], Bc = pb[b];
            const v2f ev = vadd_conj(A, Bc), d = vsub_conj(A, Bc);   // the 1/2 of Ev/Od is folded into the window
            const v2f wo = vmul_negi(twp[b], d);
            const v2f xa = ev + wo, xb = ev - wo;
            // k = 0: ev = (Re z0, 0), wo = (Im z0, 0)  ->  |xa| = |X[0]|, |xb| = |X[M]|  (same formulas)
            mg[k * MAGSTR] = __builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y);
            mg[(M - k) * MAGSTR] = __builtin_amdgcn_sqrtf(xb.x * xb.x + xb.y * xb.y);
        }
        if (t == 0) mg[(M / 2) * MAGSTR] = __builtin_amdgcn_sqrtf(zh.x * zh.x + zh.y * zh.y);
        wave_lds_fence();     // the slot's exchange image is reused by its next frame

        if (fi == FPS - 1) {
            // ---- group complete: store with lanes along the frame axis; thread = (frame column, bin row).
            // Addresses are (uniform row-block base, computed on the scalar unit) + (per-thread constant offset):
            // no per-element vector address arithmetic.  With 16 frame columns a 32-lane LDS read group spans two
            // image rows; rows 16 apart sit 16 banks apart (MAGSTR = 17), so the pairing below is conflict-free.
            __syncthreads();
            constexpr int ROWS = NT / FPB;
            const int fr = tid % FPB, rid = tid / FPB;
            const long fglob = (long)g * FPB + fr;
            float *gbase = oclip + (long)g * FPB;                      // wave-uniform
            if constexpr (FPB == 16 && ROWS >= 2 && ROWS <= 32) {
                constexpr int HALF = ROWS / 2, SUB = 16 / HALF, NBLK = (M + 1 + 31) / 32;
                const int krow0 = (rid >> 1) + 16 * (rid & 1);
                const unsigned voff = (unsigned)krow0 * (unsigned)row_stride + (unsigned)fr;
                const float *mp0 = s_mag + krow0 * MAGSTR + fr;
                if (fglob < n_frames) {
#pragma unroll
                    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
                        for (int j = 0; j < SUB; ++j) {
                            const int kk = 32 * blk + HALF * j;         // compile-time part of the row index
                            if ((kk + 31 <= M || kk + krow0 <= M) && (!FIT || kk + krow0 < out_rows))
                                (gbase + (long)kk * row_stride)[voff] = FIT ? stft_emit(mp0[kk * MAGSTR], o.quantize) : mp0[kk * MAGSTR];
                        }
                }
            } else {
                if (fglob < n_frames) {
                    float *op = oclip + (long)rid * row_stride + fglob;
                    const float *mp = s_mag + rid * MAGSTR + fr;
                    const long ostep = (long)ROWS * row_stride;
                    const int kend = M + 1 < out_rows ? M + 1 : out_rows;
#pragma unroll 4
                    for (int k = rid; k < kend; k += ROWS) {
                        *op = FIT ? stft_emit(*mp, o.quantize) : *mp;
                        op += ostep;
                        mp += ROWS * MAGSTR;
                    }
                }
            }
            __syncthreads();   // image is rewritten by the next group
        }
        fi = fn;
        g = gn;
    };
#pragma unroll 1                                        // (the explicitly unrolled form -- shift becomes renaming -- measured no faster: 4.73 ms both)
    for (int it = 0; it < n_seq; ++it) frame(it);
}

// ------------------------------------------------------------------------------------------------
// Fused STFT -> network input (adn_stft_mag_fit): PERSISTENT workgroups on whole output lines.
//
// The plain layout (bins x n_frames, n_frames odd) gives no frame group whole cache lines, which is what rules persistence
// out for stft_wave_kernel (two halves of a line must reach the L2 together).  The fitted window does not have that problem:
// rows are W floats apart (1024 bytes for 513x256), so a group of 32 frames is exactly one 128-byte line per bin row.  Here
//   * a workgroup walks a contiguous run of (clip, 32-frame group) items: the constant set-up (table copy, two barriers, 26
//     LDS reads per lane), the launch and the first frame's unprefetched HBM round trip are paid once per workgroup instead
//     of once per 16 frames, and the next group's first frames are prefetched under the last FFTs of the current one;
//   * the values leave as fp16 round trips anyway (data_loader.py:41-42), so the [bin][frame] image holds HALFS: 32 frames
//     fit the 33 KB the 16-frame fp32 image took, three workgroups per CU as before;
//   * a lane stores 16 bytes (4 frames of one row), 8 lanes one whole line: 4x fewer store instructions.
// Image: row k = 32 halfs (64 bytes); the 8-byte piece p (frames 4p .. 4p+3) of row k sits at piece p ^ ((k >> 1) & 7), so
// that the transposed writes of a frame (64 bins, one half each) spread over 16 banks (2-way, free for 2-byte writes)
// while the store phase's ds_read_b64 of 4 rows x 8 pieces per 32-lane group stays conflict-free.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int fit_col(int k, int c) { return ((((c >> 2) ^ (k >> 1)) & 7) << 2) | (c & 3); }

template <int M, int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void stft_fit_kernel(const float *__restrict__ audio, long L, int hop, int pad,
                                                               int nfc, int gpc, int total, const float *__restrict__ tables,
                                                               float *__restrict__ out, const StftOut o)
{
    constexpr int N = 2 * M, TPF = M / 8, NT = NW * 64, FPB = 32;
    constexpr int SLOTS = NT / TPF, FPS = FPB / SLOTS;       // frames per slot and group, processed in sequence
    constexpr int SCSZ = exch_size<M>();
    constexpr int R2 = WavePlan<M>::R2, R3 = WavePlan<M>::R3;
    constexpr int TBL = N + 2 * M + (M + 2);
    static_assert(TPF <= 64 && FPB % SLOTS == 0 && FPS >= 1 && M >= 128, "bad STFT tiling");
    static_assert(TBL * 4 <= (M + 1) * FPB * 2, "table must fit the magnitude image");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16 *s_img = reinterpret_cast<_Float16 *>(smem);                                   // (M+1) x 32 halfs
    v2f *s_sc = reinterpret_cast<v2f *>(smem + (((M + 1) * FPB / 2 + 3) & ~3));

    const int tid = threadIdx.x;
    const int slot = (TPF == 64) ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / TPF;
    const int t = tid - slot * TPF;
    // contiguous run of items: consecutive groups of a clip share 3/4 of a frame of audio and write neighbouring lines
    const int it0 = (int)((long)blockIdx.x * total / gridDim.x), it1 = (int)((long)(blockIdx.x + 1) * total / gridDim.x);
    if (it0 >= it1) return;
    v2f *sc = s_sc + slot * SCSZ;
    const int Li = (int)L;

    // 8-byte loads need an even sample offset from an 8-byte aligned clip (clips of odd length alternate)
    auto load_frame = [&](const float *aud, int fidx, v2f *dst) {
        const int fstart = fidx * hop - pad;
        const float *ap = aud + fstart + 2 * t;
        if (fstart >= 0 && fstart + N <= Li && !(reinterpret_cast<uintptr_t>(aud + fstart) & 7)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[u] = *reinterpret_cast<const v2f *>(ap + 2 * u * TPF);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = fstart + 2 * (t + u * TPF);
                dst[u].x = (s >= 0 && s < Li) ? ap[2 * u * TPF] : 0.f;
                dst[u].y = (s + 1 >= 0 && s + 1 < Li) ? ap[2 * u * TPF + 1] : 0.f;
            }
        }
    };

    int clip = it0 / gpc, g = it0 - clip * gpc, fi = 0;
    const float *aud = audio + (long)clip * L;
    v2f nx[8];
    load_frame(aud, g * FPB + slot * FPS, nx);            // requested before the constant set-up (overlaps it)

    {
        float *tb = smem;
        for (int i = tid * 4; i < TBL; i += NT * 4) {
            if (i + 4 <= TBL) *reinterpret_cast<f4 *>(tb + i) = *reinterpret_cast<const f4 *>(tables + i);
            else
                for (int j = i; j < TBL; ++j) tb[j] = tables[j];
        }
        __syncthreads();
    }
    v2f win[8], tw2[(R2 - 1) * (8 / R2)], tw3[R3 > 1 ? (R3 - 1) * (8 / R3) : 1], twp[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) win[u] = 0.5f * *reinterpret_cast<const v2f *>(smem + 2 * (t + u * TPF));
    load_pass_twiddles<M, R2, 8>(smem, t, tw2);
    if constexpr (R3 > 1) load_pass_twiddles<M, R3, 8 * R2>(smem, t, tw3);
    {
        const v2f *g2 = reinterpret_cast<const v2f *>(smem + N + 2 * M);
#pragma unroll
        for (int b = 0; b < 4; ++b) twp[b] = g2[t + b * TPF];
    }
    __syncthreads();

    const bool vec_ok = ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && !(o.row_stride & 3) && !(o.clip_stride & 3);
    const int n_seq = (it1 - it0) * FPS;
#pragma unroll 1
    for (int it = 0; it < n_seq; ++it) {
        const int fcol = slot * FPS + fi;
        v2f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = nx[u] * win[u];
        // next frame of this slot: the next column, or the first column of the next item (possibly the next clip)
        int fn = fi + 1, gn = g, cn = clip;
        if (fn == FPS) {
            fn = 0;
            if (++gn == gpc) { gn = 0; ++cn; }
        }
        if (it + 1 < n_seq) {
            const float *an = audio + (long)cn * L;
            const int fnx = gn * FPB + slot * FPS + fn;
            const int fs = fnx * hop - pad;
            if (hop * 4 == N && fn != 0 && fs >= 0 && fs + N <= Li && !(reinterpret_cast<uintptr_t>(an + fs) & 7)) {
#pragma unroll
                for (int u = 0; u < 6; ++u) nx[u] = nx[u + 2];
                const float *ap = an + fs + 2 * t;
                nx[6] = *reinterpret_cast<const v2f *>(ap + 2 * 6 * TPF);
                nx[7] = *reinterpret_cast<const v2f *>(ap + 2 * 7 * TPF);
            } else {
                load_frame(an, fnx, nx);
            }
        }

        vdft<8>(v);
#pragma unroll
        for (int q = 0; q < 8; ++q) ADN_XWR(sc + xidx<M, 1>(8 * t + q), v[q]);
        wave_lds_fence();
        wave_pass<M, R2, 8, 1, (R3 > 1 ? 2 : 3)>(sc, tw2, t, v);
        if constexpr (R3 > 1) wave_pass<M, R3, 8 * R2, 2, 3>(sc, tw3, t, v);

        v2f pa[4], pb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = t + b * TPF;
            pa[b] = ADN_XRD(sc + xidx<M, 3>(k));
            pb[b] = ADN_XRD(sc + xidx<M, 3>((M - k) & (M - 1)));
        }
        const v2f zh = 2.0f * ADN_XRD(sc + xidx<M, 3>(M / 2));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = t + b * TPF;
            const v2f A = pa[b], Bc = pb[b];
            const v2f ev = vadd_conj(A, Bc), d = vsub_conj(A, Bc);
            const v2f wo = vmul_negi(twp[b], d);
            const v2f xa = ev + wo, xb = ev - wo;
            s_img[k * FPB + fit_col(k, fcol)] = (_Float16)__builtin_amdgcn_sqrtf(xa.x * xa.x + xa.y * xa.y);
            s_img[(M - k) * FPB + fit_col(M - k, fcol)] = (_Float16)__builtin_amdgcn_sqrtf(xb.x * xb.x + xb.y * xb.y);
        }
        if (t == 0) s_img[(M / 2) * FPB + fit_col(M / 2, fcol)] = (_Float16)__builtin_amdgcn_sqrtf(zh.x * zh.x + zh.y * zh.y);
        wave_lds_fence();

        if (fi == FPS - 1) {
            // ---- group complete: 8 lanes per row, 16 bytes (4 frames) per lane -> whole 128-byte lines ----
            __syncthreads();
            const int p = tid & 7, f0 = g * FPB + 4 * p;
            float *ob = out + (long)clip * o.clip_stride + f0;
            if (f0 < nfc) {
#pragma unroll 4
                for (int row = tid >> 3; row < o.rows; row += NT / 8) {
                    const h4 hv = *reinterpret_cast<const h4 *>(s_img + row * FPB + (((p ^ (row >> 1)) & 7) << 2));
                    const f4 fv = {(float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w};
                    float *op = ob + (long)row * o.row_stride;
                    if (vec_ok && f0 + 4 <= nfc) {
                        *reinterpret_cast<f4 *>(op) = fv;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (f0 + j < nfc) op[j] = fv[j];
                    }
                }
            }
            __syncthreads();   // image is rewritten by the next group
        }
        fi = fn;
        g = gn;
        clip = cn;
    }
}

template <int M, int NW, int FPB, int WPE = 3>
hipError_t launch_wave(const float *audio, int n_clips, long L, int hop, int pad, long n_frames, const float *tables,
                       float *out, const StftOut &o, hipStream_t st, int gpb)
{
    constexpr int TPF = M / 8, SLOTS = NW * 64 / TPF;
    const long groups = (n_frames + FPB - 1) / FPB;
    if (gpb < 1) gpb = 1;
    if (gpb > groups) gpb = (int)groups;
    const long bpc = (groups + gpb - 1) / gpb;
    const long nwg = bpc * n_clips;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = (size_t)((((M + 1) * (FPB + 1) + 1) & ~1) + 2 * SLOTS * exch_size<M>()) * sizeof(float);
    const bool fit = o.quantize != 0;
    auto kern = fit ? stft_wave_kernel<M, NW, FPB, WPE, true> : stft_wave_kernel<M, NW, FPB, WPE, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, st, audio, L, hop, pad, n_frames, (int)groups,
                       gpb, (int)bpc, tables, out, o);
    return hipGetLastError();
}

template <int M, int NW, int WPE = 3>
hipError_t launch_fit(const float *audio, int n_clips, long L, int hop, int pad, long n_frames, const float *tables,
                      float *out, const StftOut &o, hipStream_t st)
{
    constexpr int TPF = M / 8, SLOTS = NW * 64 / TPF, FPB = 32;
    const long gpc = (n_frames + FPB - 1) / FPB;
    const long total = gpc * n_clips;
    if (total <= 0 || total > 0x7fffffffL || n_frames > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = (size_t)((((M + 1) * FPB / 2 + 3) & ~3) + 2 * SLOTS * exch_size<M>()) * sizeof(float);
    auto kern = stft_fit_kernel<M, NW, WPE>;
    // persistent grid: exactly as many workgroups as the chip holds at once (asked of the runtime for THIS kernel and LDS size)
    struct Res { int dev = -1, wgs = 0; };
    static thread_local Res cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (cache.dev != dev) {
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        int per_cu = 0, cus = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NW * 64, lds);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (per_cu < 1 || cus < 1) return hipErrorInvalidValue;
        cache.dev = dev;
        cache.wgs = per_cu * cus;
    }
    const long nwg = total < cache.wgs ? total : cache.wgs;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, st, audio, L, hop, pad, (int)n_frames, (int)gpc,
                       (int)total, tables, out, o);
    return hipGetLastError();
}

struct TableKey {
    int device, n_fft;
    bool operator<(const TableKey &o) const { return device != o.device ? device < o.device : n_fft < o.n_fft; }
};
std::mutex g_table_mu;
std::map<TableKey, float *> g_tables;

hipError_t get_tables(int n_fft, const float **out, hipStream_t st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_table_mu);
    auto it = g_tables.find(TableKey{dev, n_fft});
    if (it != g_tables.end()) { *out = it->second; return hipSuccess; }
    if (stream_is_capturing(st)) return ADN_COLD_IN_CAPTURE;      // the upload below blocks: not inside a capture (adn_prepare)
    const int N = n_fft, M = N / 2;
    std::vector<float> h((size_t)N + 2 * M + (M + 2), 0.f);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < N; ++i) h[i] = (float)(0.5 - 0.5 * std::cos(2.0 * pi * i / N));   // periodic Hann
    for (int j = 0; j < M; ++j) {
        h[N + 2 * j] = (float)std::cos(2.0 * pi * j / M);
        h[N + 2 * j + 1] = (float)(-std::sin(2.0 * pi * j / M));
    }
    for (int k = 0; k <= M / 2; ++k) {
        h[N + 2 * M + 2 * k] = (float)std::cos(2.0 * pi * k / N);
        h[N + 2 * M + 2 * k + 1] = (float)(-std::sin(2.0 * pi * k / N));
    }
    float *d = nullptr;
    e = hipMalloc(&d, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    g_tables[TableKey{dev, n_fft}] = d;
    *out = d;
    return hipSuccess;
}

template <int M>
hipError_t launch_m(const float *audio, int n_clips, long L, int hop, int pad, long n_frames, const float *tables,
                    float *out, const StftOut &o, hipStream_t st)
{
    using C = StftCfg<M>;
    const long groups = (n_frames + C::FPB - 1) / C::FPB;
    const long nwg = groups * n_clips;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    const long span = (long)(C::FB - 1) * hop + C::N;
    const long scratch = (long)C::FB * M * 2;
    const long work = span > scratch ? span : scratch;
    const long mag = (long)(M + 1) * C::MAGSTR;
    const size_t lds = (size_t)(C::N + 2 * M + (M + 2) + mag + (mag & 1) + work) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;   // hop too large for the LDS staging scheme
    auto kern = stft_mag_kernel<M>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(STFT_THREADS), lds, st, audio, L, hop, pad, n_frames,
                       (int)groups, tables, out, (int)work, o);
    return hipGetLastError();
}

}  // namespace

// window / twiddle tables of one n_fft on the current device (cached): win[N], tw[M] = exp(-2 pi i j / M),
// tw2[M/2+1] = exp(-2 pi i k / N); shared with the inverse-STFT kernels
hipError_t stft_tables(int n_fft, const float **out, hipStream_t st) { return get_tables(n_fft, out, st); }

bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// n_frames = frames to compute per clip (all of them, or only those inside the fitted window); rows / row_stride /
// clip_stride / quantize: see StftOut
hipError_t launch_stft_mag(const float *audio, int n_clips, long L, int n_fft, int hop, int center, long n_frames,
                           float *out, int rows, long row_stride, long clip_stride, int quantize, hipStream_t st)
{
    const StftOut o{row_stride, clip_stride, rows, quantize};
    const float *tables = nullptr;
    hipError_t e = get_tables(n_fft, &tables, st);
    if (e != hipSuccess) return e;
    const int pad = center ? n_fft / 2 : 0;
    constexpr int gpb = 1;                                // frame groups per workgroup (1 measured best: no in-loop barriers)
    if (quantize) {                                       // adn_stft_mag_fit: persistent whole-line kernel (above)
        switch (n_fft) {
            case 256: return launch_fit<128, 4>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st);
            case 512: return launch_fit<256, 4>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st);
            case 1024: return launch_fit<512, 4>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st);
            default: break;
        }
    }
    switch (n_fft) {                                      // a frame fits one wave: the wave-synchronous kernel
        case 64: return launch_wave<32, 1, 16>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st, gpb);
        case 128: return launch_wave<64, 2, 16>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st, gpb);
        case 256: return launch_wave<128, 4, 16>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st, gpb);
        // <waves, frames per group>, measured on 10 k x 3 s clips (ms): <8,16> 9.31, <4,16> 5.42, <2,32> 6.08, <2,16> 4.78, <4,32> 4.71
        case 512: return launch_wave<256, 4, 32>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st, gpb);
        case 1024: return launch_wave<512, 4, 16>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st, gpb);
        // larger transforms: the workgroup-synchronous kernel
        case 2048: return launch_m<1024>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st);
        case 4096: return launch_m<2048>(audio, n_clips, L, hop, pad, n_frames, tables, out, o, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adn
