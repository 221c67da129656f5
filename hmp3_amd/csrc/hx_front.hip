// hx_front.hip - front-end kernels of the batched MP3 encoder for MI355X (gfx950):
//   k_dcfilter   K0  optional input DC blocker, sequential per channel   (filter2.c:116-144)
//   k_polyphase  K1  int16 / fp32 PCM -> 32 x 18 subband samples per granule (the stage of sbt.c:57-310), and in its
//                    epilogue the subband energies in mB for the transient detector (detect.c:80-101)
//                    (its first tile of a stream also forms the energies of the carried granule: the call's first detector index)
//   k_detect     K2  attack metric for both "previous granule short" cases (detect.c:103-141) and the per-stream block-type
//                    state machine (mp3enc.cpp:1398-1440), one wavefront per stream
//   k_spec       K4  window + 18-point (3 x 6-point) MDCT + alias butterflies (hwin.c:147-322,
//                    emdct.c:104-288), MDCT-energy psy model (emap.c:61-96, spdsmr.c:64-273) and the
//                    L/R vs M/S metric (bitallo3.cpp:682-742, bitallos.cpp:377-416), one wave per granule
//   k_msscan     K5a the frames' stereo decisions: hysteresis scan over a stream's granules (bitallo3.cpp:693-751)
//   k_prep       K5b what the allocator's granule start needs that does not depend on its carried state: signs, band
//                    energies, band maxima of x^(3/4), zero-gain steps, masks (bitallo3.cpp:816-1066, spdsmr.c:275-318)
//                    (k_msscan also rolls the 3-granule subband carry and the PCM history into the next call)
// Parallel over streams x channels x granules (x slots / subbands / partitions).  Each lane
// evaluates its unit with the reference's operation order, so results are bit-identical.
// The file is compiled twice (hmp3_amd/build.sh): HX_FRONT_PART=1 holds k_polyphase and the small kernels, HX_FRONT_PART=2
// k_spec / k_msscan / k_prep - the latter with LLVM's iterative-ilp scheduling strategy, which suits their long
// straight-line arithmetic (k_prep 1.50 -> 1.36 ms) and hurts the polyphase loop (1.17 -> 1.38 ms).
#include "hx_dev.h"
#ifndef HX_FRONT_PART
#define HX_FRONT_PART 3     // both
#endif

#if HX_FRONT_PART & 1
// (K1_GPB granules per workgroup, K1_THREADS lanes: hx_types.h, shared with the launch in hx_cabi.hip)

#define K1_NS (480 + 576 * K1_GPB)      // staged samples
#define K1_LDS (K1_NS + (K1_NS >> 5) + 1)

// N-point DCT of the analysis filterbank by odd / even decimation:
//   X[j], X[N-1-j] = E[j] +- tw_N[j] * O[j]   with   E = DCT_{N/2}(x[0], x[2], ...),  O = DCT_{N/2}(u),
//   u[i] = x[2i+1] - u[i+1],  u[N/2-1] = x[N-1]   (running difference of the odd samples from the top)
// tw_N[j] = 2 cos(pi (2j+1) / 2N) sits at tw[N/2 + j].  Everything stays in registers: the recursion is
// resolved at compile time into straight-line code, depth first.  Each sum is one rounding, in the order
// written here, which is also the reference's (sbt.c:134-259), so the subband samples agree bit for bit.
template <int N> struct AnalysisDct {
    template <class T>
    static __device__ __forceinline__ void run(const T *x, T *X, const float *tw)
    {
        constexpr int H = N / 2;
        T even[H], u[H], E[H], O[H];
        u[H - 1] = x[N - 1];
        even[H - 1] = x[N - 2];
#pragma unroll
        for (int i = H - 2; i >= 0; i--) { u[i] = x[2 * i + 1] - u[i + 1]; even[i] = x[2 * i]; }
        AnalysisDct<H>::run(even, E, tw);
        AnalysisDct<H>::run(u, O, tw);
#pragma unroll
        for (int j = 0; j < H; j++) {
            const T r = tw[H + j] * O[j];
            X[j] = E[j] + r;
            X[N - 1 - j] = E[j] - r;
        }
    }
};
template <> struct AnalysisDct<1> {
    template <class T>
    static __device__ __forceinline__ void run(const T *x, T *X, const float *) { X[0] = x[0]; }
};

// One lane = one time slot of BOTH channels: every value is a (left, right) pair and the arithmetic is packed fp32
// (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and instruction, each rounded like the plain one), so the
// window taps, the LDS reads and the instruction stream are shared by the two channels.  A workgroup of four waves
// stages the interleaved stereo PCM of K1_GPB granules (plus 480 samples of history) as float pairs; a sample pair is
// one ds_read_b64 (2 LDS cycles per wave; stride 33 pairs between lanes = 66 words: conflict-free), the taps come
// through the scalar cache.  Round 3: 1.165 -> 0.865 ms per 1024 x 256 frames against one channel per lane with
// ds_read2_b32 samples and taps by LDS broadcast; the kernel now moves 3.6 GB in that time and is bound by HBM.
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(K1_THREADS) void k_polyphase(const int16_t *__restrict__ pcm, long long nsamp,
                                                   const HxStream *__restrict__ st,
                                                   const HxParams *__restrict__ prm,
                                                   const HxGlobalTabs *__restrict__ gt,
                                                   float *__restrict__ sb, int NG, int SG,
                                                   const float *__restrict__ pcmf, int nchan, int *__restrict__ eng, int lsf)
{
    __shared__ __attribute__((aligned(16))) v2f xs[K1_LDS];
    const int s = blockIdx.x, lt = threadIdx.x;
    const int g0 = blockIdx.y * K1_GPB;
    const int ng = min(K1_GPB, NG - g0);
    const int count = 480 + 576 * ng;
    const HxStream *ss = st + s;
    const HxParams *p = prm + __builtin_amdgcn_readfirstlane(ss->cls);
    const int16_t *src = pcm + (long long) s * nsamp * nchan;   // interleaved L R (or one channel)
    if (blockIdx.y == 0 && lt < 18) {
        // The detector energies of eng index 0 (this kernel's epilogue writes the others): from the carried last granule of the
        // previous call, subband slot 2, which no workgroup of this launch writes.  (A kernel of its own until round 6: eighteen
        // lanes per stream, one launch less per call.)
        const int ch = lt / 9, k = lt - 9 * ch;
        const int sb0 = lsf ? 8 : 4, nsbb = lsf ? 20 : 14;
        const float *y = sb + ((long long) (s * 2 + ch) * SG + 2) * 576 + 18 * sb0 + 2 * k;
        float sum = 7.0e4f;
        for (int i = 0; i < nsbb; i++, y += 18) {
            float x = y[0] * y[0]; sum += x;
            x = y[1] * y[1]; sum += x;
        }
        eng[(long long) (s * 2 + ch) * NG * 9 + k] = hx_mblog(gt->mblog, sum);
    }
    const long long n0 = 576LL * g0 - 480;                      // sample index of staged slot 0
    const int hist = (g0 == 0) ? 480 : 0;                       // slots that come from the carry
    if (nchan == 1) {       // mono batch: the right half of every pair is silence
        const float *srcf = pcmf + (long long) s * nsamp;
        for (int idx = hist + lt; idx < count; idx += K1_THREADS)
            xs[idx + (idx >> 5)] = v2f{pcmf ? srcf[n0 + idx] : (float) src[n0 + idx], 0.0f};
    } else if (pcmf) {      // DC-blocked input from k_dcfilter: fp32, interleaved like the PCM
        const v2f *srcf = reinterpret_cast<const v2f *>(pcmf) + (long long) s * nsamp;
        for (int idx = hist + lt; idx < count; idx += K1_THREADS) xs[idx + (idx >> 5)] = srcf[n0 + idx];
    } else if ((reinterpret_cast<unsigned long long>(pcm) & 15ull) == 0) {
        const int nv = count >> 2, vh = hist >> 2;              // 4 stereo samples per 16 bytes
        constexpr int NR = ((480 + 576 * K1_GPB) / 4 + K1_THREADS - 1) / K1_THREADS;
        int4 w[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int v = lt + K1_THREADS * r;
            const int vc = min(max(v, vh), nv - 1);
            w[r] = *reinterpret_cast<const int4 *>(src + 2 * (n0 + 4LL * vc));
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int v = lt + K1_THREADS * r;
            if (v >= vh && v < nv) {
                const int q[4] = {w[r].x, w[r].y, w[r].z, w[r].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int idx = 4 * v + e;
                    xs[idx + (idx >> 5)] = v2f{(float) (short) (q[e] & 0xFFFF), (float) (short) (q[e] >> 16)};
                }
            }
        }
    } else {
        for (int idx = hist + lt; idx < count; idx += K1_THREADS) {
            const long long n = n0 + idx;
            xs[idx + (idx >> 5)] = v2f{(float) src[2 * n], (float) src[2 * n + 1]};
        }
    }
    for (int i = lt; i < hist; i += K1_THREADS) xs[i + (i >> 5)] = v2f{ss->pcm_hist[0][i], nchan == 2 ? ss->pcm_hist[1][i] : 0.0f};
    __syncthreads();
    const int gl = lt / 18, t = lt - gl * 18;
    if (gl >= ng) return;
    const int base = 480 + 576 * gl + 32 * t + 31;          // newest sample of the slot
    // (volatile: single ds_read_b64, 2 LDS cycles each; merged into ds_read2_b64 a pair would take 8)
    typedef const volatile v2f __attribute__((address_space(3))) *LdsPairPtr;
    LdsPairPtr P = (LdsPairPtr) (xs + (base + (base >> 5) - 526));      // P[526 - pad(off)] = sample of age off
#define XS(off) P[526 - ((off) + ((off) >> 5))]
    // Two-stage pipeline over the 32 folded window lines, pinned with scheduling barriers: the 16 sample pairs of line
    // k + 1 are in flight while line k is summed.
    v2f b[32], X[32];
    v2f xa[2][8], xb[2][8];
    // The taps are the same for every lane: scalar loads into SGPRs, two lines ahead (a scalar load returns out of order,
    // so waiting for one means lgkmcnt(0): that wait stands at the top of a step, where the step's sample pairs are due
    // anyway, and the next line's reads are issued behind it).
    const float *wg = gt->anwin_r;
    float w[3][16];
#define K1_WLOAD(k) { _Pragma("unroll") for (int i = 0; i < 16; i++) w[(k) % 3][i] = wg[16 * (k) + i]; }
#define K1_TAP(k, i) w[(k) % 3][i]
#define K1_LOAD(k) { \
        const int A_ = ((k) == 0) ? 16 : ((k) <= 16) ? 16 + (k) : 80 - (k); \
        const int B_ = ((k) <= 16) ? 16 - (k) : 16 + (k); \
        _Pragma("unroll") for (int j = 0; j < 4; j++) { \
            xa[(k) & 1][2 * j] = XS(A_ + 128 * j); \
            xa[(k) & 1][2 * j + 1] = XS(A_ + 128 * j + 64); \
            if ((k) != 0) { xb[(k) & 1][2 * j] = XS(B_ + 128 * j); xb[(k) & 1][2 * j + 1] = XS(B_ + 128 * j + 64); } \
        } }
    K1_LOAD(0)
    K1_WLOAD(0)
    K1_WLOAD(1)
#pragma unroll
    for (int k = 0; k < 32; k++) {
        __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (k + 2 < 32) K1_WLOAD(k + 2)
        if (k + 1 < 32) K1_LOAD(k + 1)
        __builtin_amdgcn_sched_barrier(0);
        v2f s1 = {0.0f, 0.0f}, s2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // (tap * pair: the packed multiply's operand selectors splat the SGPR over both channels)
            const float c0x = K1_TAP(k, 4 * j), c0y = K1_TAP(k, 4 * j + 1), c1x = K1_TAP(k, 4 * j + 2), c1y = K1_TAP(k, 4 * j + 3);
            s1 += c0x * xa[k & 1][2 * j];
            if (k) s2 += c0y * xb[k & 1][2 * j];
            s1 += c1x * xa[k & 1][2 * j + 1];
            if (k) s2 += c1y * xb[k & 1][2 * j + 1];
        }
        b[k] = k ? s1 + s2 : s1;
        __builtin_amdgcn_sched_barrier(0);
    }
#undef K1_LOAD
#undef K1_WLOAD
#undef K1_TAP
#undef XS
    AnalysisDct<32>::run(b, X, p->dct_tw);
    float *out = sb + ((long long) (s * 2) * SG + (g0 + gl + 3)) * 576 + t;
#ifdef HX_MOCK_NOSB_STORE
    // (timing experiment, EXPERIMENTS.md round 6: what the polyphase costs when its output does not go to HBM - the bound of a
    // polyphase -> MDCT fusion; the condition is false at run time, the compiler cannot know)
    if (NG > 0x7fff0000)
#endif
    {
#pragma unroll
    for (int k = 0; k < 32; k++) out[18 * k] = X[k].x;
    }
#ifdef HX_MOCK_NOSB_STORE
    if (NG > 0x7fff0000)
#endif
    if (nchan == 2) {
        float *out1 = out + (long long) SG * 576;
#pragma unroll
        for (int k = 0; k < 32; k++) out1[18 * k] = X[k].y;
    }
    {   // transient detector input (reference detect.c:147-196): energy of subbands 4..17 (MPEG-2 LSF rates: 8..27) per pair
        // of time slots, as mB: slots 2 k (this lane) and 2 k + 1 (the next lane), subband after subband, in the reference's
        // order of additions.  eng index g <-> the granule one before coded granule g, so this granule's go to index + 1
        // (the last granule's are next call's index 0, formed from the carry at the top of this kernel).  A mono batch's silent
        // second channel sums to the floor by itself.
        v2f sum = {7.0e4f, 7.0e4f};
#define ENG_TERM(i) { const v2f y1 = {__shfl_down(X[i].x, 1, 64), __shfl_down(X[i].y, 1, 64)}; v2f x = X[i] * X[i]; sum += x; x = y1 * y1; sum += x; }
        if (!lsf) {
#pragma unroll
            for (int i = 4; i < 18; i++) ENG_TERM(i)
        } else {
#pragma unroll
            for (int i = 8; i < 28; i++) ENG_TERM(i)
        }
#undef ENG_TERM
        if (!(t & 1) && g0 + gl + 1 < NG) {
            int *eo = eng + ((long long) (s * 2) * NG + g0 + gl + 1) * 9 + (t >> 1);
            eo[0] = hx_mblog(gt->mblog, sum.x);
            eo[(long long) NG * 9] = hx_mblog(gt->mblog, sum.y);
        }
    }
}

// attack metric of one channel at coded step g, for short_flag_prev = 0 and 1
// (the MPEG-2 detector looks back four values instead of six, detect.c:205-226)
__device__ __forceinline__ void attack_metric(const int *hist, const int *eng, int g, int *m0, int *m1, int lsf)
{
    // virtual buffer A: 32 history values followed by 9 new values per step
    int w[32];
#pragma unroll
    for (int j = 10; j < 29; j++) {
        int a = 9 * (g + 1) + j;
        w[j] = (a < 32) ? hist[a] : eng[a - 32];
    }
    int r0 = 0, r1 = 0;
#pragma unroll
    for (int j = 17; j < 29; j++) {
        int a0 = lsf ? -0x7fffffff : max(w[j - 6], w[j - 7]);
        int a1 = max(w[j - 4], w[j - 5]);
        int a2 = max(w[j - 2], w[j - 3]);
        a1 = max(a1, a0);
        int a = max(a1, a2);
        int d = w[j] - a;
        r0 = max(r0, d);
        if (j >= 18) r1 = max(r1, d);
    }
    *m0 = r0;
    *m1 = r1;
}

// Transient flags and block types of a stream, one wavefront per stream (round 6: two kernels before - a flag per (stream,
// granule) lane, then one lane per stream walking them): 64 granules at a time the lanes form their granule's two flags (for
// short_flag_prev = 0 and 1), lane 0 walks the state machine over them - block_type[g] = table[prev type][short now][short
// next] - and the lanes store the 64 types.
__global__ __launch_bounds__(256) void k_detect(HxStream *__restrict__ st, const HxParams *__restrict__ prm,
                                                const int *__restrict__ eng, unsigned char *__restrict__ flg,
                                                int *__restrict__ dbg_metric, unsigned char *__restrict__ bt,
                                                unsigned char *__restrict__ btprev, int NG, int S, int lsf)
{
    __shared__ unsigned char sflg[4][64], sbt[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wv;
    if (s >= S) return;         // (wave-uniform; nothing below is a workgroup barrier)
    HxStream *ss = st + s;
    const int thr = prm[ss->cls].short_block_threshold;
    // sel[prev type * 4 + short now * 2 + short next] = {0, 1, 2, 2, 3, 2, 2, 2, 3, 2, 2, 2, 0, 1, 2, 2} as nibbles of a constant
    const unsigned long long sel = 0x2210222322232210ull;
    int prev_next = ss->short_flag_next_prev, prev_bt = ss->bt_prev;
    if (lane == 0) btprev[s] = (unsigned char) prev_bt;
    const int *e0 = eng + (long long) (s * 2 + 0) * NG * 9, *e1 = eng + (long long) (s * 2 + 1) * NG * 9;
    for (int gb = 0; gb < NG; gb += 64) {
        const int g = gb + lane;
        if (g < NG) {
            int a0, a1, b0, b1;
            attack_metric(ss->attack_hist[0], e0, g, &a0, &a1, lsf);
            attack_metric(ss->attack_hist[1], e1, g, &b0, &b1, lsf);
            const int f0 = (a0 > thr) | (b0 > thr), f1 = (a1 > thr) | (b1 > thr);
            const unsigned char f = (unsigned char) (f0 | (f1 << 1));
            sflg[wv][lane] = f;
            flg[(long long) s * NG + g] = f;
            if (dbg_metric) { dbg_metric[((long long) s * NG + g) * 2] = a0; dbg_metric[((long long) s * NG + g) * 2 + 1] = b0; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int n = min(64, NG - gb);
            for (int k = 0; k < n; k++) {
                const int f = sflg[wv][k];
                const int next = prev_next ? (f >> 1) & 1 : f & 1;
                const int b = (int) ((sel >> (4 * (prev_bt * 4 + prev_next * 2 + next))) & 15);
                prev_bt = b;
                prev_next = next;
                sbt[wv][k] = (unsigned char) b;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (g < NG) bt[(long long) s * NG + g] = sbt[wv][lane];
        __builtin_amdgcn_wave_barrier();
    }
    // roll the energy history: last 32 values of [hist | eng] (every lane has read the history above)
    int keep = 0;
    const int c = lane >> 5, j = lane & 31;
    {
        const int *e = c ? e1 : e0;
        const int a = 9 * NG + j;
        keep = (a < 32) ? ss->attack_hist[c][a] : e[a - 32];
    }
    __builtin_amdgcn_wave_barrier();
    ss->attack_hist[c][j] = keep;
    if (lane == 0) { ss->short_flag_next_prev = prev_next; ss->bt_prev = prev_bt; }
}
#endif      // HX_FRONT_PART & 1
#if HX_FRONT_PART & 2
// ---- MDCT kernels ------------------------------------------------------------------------------------
// An N-point kernel (N = 18 for long blocks, 6 for each short window) maps the folded, windowed input f to
// N spectral lines in three steps:
//   1. twiddle and pair the inputs:  g_i = pre[i] f[i];  s_i = g_i + g_{N-1-i};  d_i = odd[i] (g_i - g_{N-1-i})
//   2. an N/2-point cosine transform C of each half:  S = C(s),  D = C(d)
//   3. un-twist:  T_0 = D_0, T_k = D_k - T_{k-1};  y = (S_0, T_0, S_1, T_1, ...) with every element after the
//      first reduced by its finished predecessor.
// The cosine transforms are written out below; sums run left to right.  Operation order equals the
// reference's (emdct.c:104-303), hence bit-identical spectra.

// ((c0 v0 + c1 v1) + c2 v2) + c3 v3
__device__ __forceinline__ float dot4(const float *c, const float *v) { return c[0] * v[0] + c[1] * v[1] + c[2] * v[2] + c[3] * v[3]; }

struct Cos9 {       // 9-point: the input is folded once more into 5 sums (-> even outputs) and 4 differences (-> odd)
    static constexpr int n = 9;
    static __device__ __forceinline__ void run(const HxParams *p, const float *u, float *X)
    {
        float e[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { e[q] = u[q] + u[8 - q]; o[q] = u[q] - u[8 - q]; }
        const float mid = u[4];
        X[0] = 0.5f * (e[0] + e[1] + e[2] + e[3] + mid);
        X[6] = 0.5f * (e[0] + e[2] + e[3]) - e[1] - mid;
        X[3] = p->dct9_k3 * (o[0] - o[2] - o[3]);
        X[2] = dot4(p->dct9_even[0], e) - mid;
        X[4] = dot4(p->dct9_even[1], e) + mid;
        X[8] = dot4(p->dct9_even[2], e) + mid;
        X[1] = dot4(p->dct9_odd[0], o);
        X[5] = dot4(p->dct9_odd[1], o);
        X[7] = dot4(p->dct9_odd[2], o);
    }
};

struct Cos3 {       // 3-point
    static constexpr int n = 3;
    static __device__ __forceinline__ void run(const HxParams *p, const float *u, float *X)
    {
        const float e = u[0] + u[2];
        X[0] = e + u[1];
        X[1] = p->dct3_k * (u[0] - u[2]);
        X[2] = e - u[1] - u[1];
    }
};

template <class Cos>
__device__ __forceinline__ void mdct_kernel(const HxParams *p, const float *pre, const float *odd, const float *f, float *y)
{
    constexpr int H = Cos::n, N = 2 * H;
    float s[H], d[H], S[H], D[H];
#pragma unroll
    for (int i = 0; i < H; i++) {
        const float lo = pre[i] * f[i], hi = pre[N - 1 - i] * f[N - 1 - i];
        s[i] = lo + hi;
        d[i] = odd[i] * (lo - hi);
    }
    Cos::run(p, s, S);
    Cos::run(p, d, D);
    float twist = D[0];
    y[0] = S[0];
    y[1] = twist - y[0];
#pragma unroll
    for (int k = 1; k < H; k++) {
        twist = D[k] - twist;
        y[2 * k] = S[k] - y[2 * k - 1];
        y[2 * k + 1] = twist - y[2 * k];
    }
}

// ---------------------------------------------------------------------------------------
// K4: hybrid MDCT + psychoacoustic model + M/S metric of one (stream, granule), one wavefront.
// The two subband blocks the transform needs (both channels) are fetched with coalesced 16-byte
// loads into LDS; the spectrum goes back to global memory the same way and stays in LDS for the
// psy model (lane = partition, channel after channel) and the M/S metric (lane = sfb), which
// therefore never re-read it from HBM.
//
// MDCT: lane = subband of channel lane >> 5.  Frequency inversion (hwin.c:282) is applied while
// reading, so the stored subband samples stay un-inverted; the alias butterflies exchange 8
// values with each neighbour lane.

// (hand-overs inside a wave need no workgroup barrier: its LDS operations execute in order)
// (and no s_waitcnt either: a wave's DS instructions are taken in issue order)
#define FE_WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// The lookup tables every psy / metric step gathers from (mB logarithm, mB exponential): staged in LDS once per
// workgroup.  Gathers from global memory queue behind the kernel's own streaming loads and stores in the vector
// memory path - a dozen dependent ones per channel were three quarters of this kernel's time.
struct SpecTabs { int mblog[256]; float mbexp_lo[256], mbexp_hi[256]; };

// Psychoacoustic model of a short granule's channel (reference emap.c:61-93, spdsmr.c:64-107): per-window partition
// energies, then mask[w][sfb] = spread(2 sfb partitions); pre-echo control happens in the allocator.  x = the
// channel's 576 lines (LDS); thr gets mask[12*w + sfb], etab zeros.  Short granules are rare: table reads from global memory.
// The function is out of line, so its pointers carry their address spaces: x and es are LDS, the rest global memory.  As
// generic pointers they compiled to FLAT loads and stores, and the wave-local hand-overs here (FE_WAVE_SYNC: no s_waitcnt,
// DS instructions execute in issue order) promise nothing about FLAT accesses that land in LDS.  tools/check_lds_flat.py
// fails the build check if a FLAT instruction reappears in a front-end or packing function.
#define HX_LDS __attribute__((address_space(3)))
#define HX_GLB __attribute__((address_space(1)))
__device__ __noinline__ void psy_short(const HX_LDS float *x, const HX_GLB HxParams *p, HX_GLB float *etab_out, HX_GLB float *thr_out, HX_LDS float (*es)[64])
{
    const int lane = threadIdx.x & 63;
    const HX_GLB HxPsyTab *ps = &p->psyS;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (lane < ps->npart_e) {
        int i0 = ps->pstart[lane], n = ps->nsum[lane];
        for (int k = 0; k < n; k++) {
            s0 += x[i0 + k] * x[i0 + k];
            s1 += x[192 + i0 + k] * x[192 + i0 + k];
            s2 += x[384 + i0 + k] * x[384 + i0 + k];
        }
    }
    es[0][lane] = s0; es[1][lane] = s1; es[2][lane] = s2;
    FE_WAVE_SYNC();
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
    if (lane < 12 && 2 * lane < ps->npart) {
        float a[3] = {0.5f, 0.5f, 0.5f}, b[3] = {0.5f, 0.5f, 0.5f};
        int i = 2 * lane, q = ps->off[i], n = ps->cnt[i], r = ps->row[i];
        for (int j = 0; j < n; j++)
            for (int w = 0; w < 3; w++) a[w] += ps->w[r + j] * es[w][q + j];
        q = ps->off[i + 1]; n = ps->cnt[i + 1]; r = ps->row[i + 1];
        for (int j = 0; j < n; j++)
            for (int w = 0; w < 3; w++) b[w] += ps->w[r + j] * es[w][q + j];
        m0 = a[0] + b[0]; m1 = a[1] + b[1]; m2 = a[2] + b[2];
    }
    etab_out[lane] = 0.0f;
    // thr layout for short granules: [12*w + sfb]
    if (lane < 12) { thr_out[lane] = m0; thr_out[12 + lane] = m1; thr_out[24 + lane] = m2; }
    else if (lane >= 36) thr_out[lane] = 0.0f;
    FE_WAVE_SYNC();
}

// Psychoacoustic model of a long granule, both channels in one pass (lane = partition): partition energies, spreading
// (reference emap.c / spdsmr.c:185-262), signal-to-noise statistics and the threshold scale.  The lane's table entries
// (pc: first line, lines, spreading row start / length / first source, absolute threshold) were read at the start of
// the kernel; the spreading weights of a row, the same for both channels, are read once for the two and eight at a
// time; per channel the order of operations is the reference's.  Outputs etab (energy + absolute threshold) and
// thr = a * stab (threshold before pre-echo control).
struct PsyLane { int i0, nsum, off, cnt, row; float wabs; };
__device__ __forceinline__ void psy_long2(const float *xl, const HxParams *p, const SpecTabs &T, const PsyLane &pc,
                                          float *etab_out, float *thr_out, float (*xtab)[64])
{
    const int lane = threadIdx.x & 63;
    const HxPsyTab *pt = &p->psyL;
    const float *w = pt->w;
    const float alpha = 0.30f;
    const int npart = pt->npart, npart2 = (npart + 1) & (~1);
    float e[2] = {0.0f, 0.0f};
    if (lane < pt->npart_e) {
        float s0 = 0.0f, s1 = 0.0f;
        const float *xa = xl + pc.i0, *xb = xl + 576 + pc.i0;
        for (int k = 0; k < pc.nsum; k++) { s0 += xa[k] * xa[k]; s1 += xb[k] * xb[k]; }
        e[0] = s0; e[1] = s1;
    }
    float et[2] = {0.0f, 0.0f};
    int mbe[2] = {0, 0};
    if (lane < npart2) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            et[c] = pc.wabs + e[c];
            mbe[c] = hx_mblog(T.mblog, et[c]);
            xtab[c][lane] = hx_mbexp(T.mbexp_lo, T.mbexp_hi, (int) (alpha * mbe[c]));
        }
    }
    FE_WAVE_SYNC();
    float sacc[2] = {0.1f, 0.1f};
    if (lane < npart) {
        const float *wr = w + pc.row, *xa = xtab[0] + pc.off, *xb = xtab[1] + pc.off;
        const int n = pc.cnt;
        // 16 weights per round, four 16-byte reads in flight (rows start at any word; what a read takes in beyond the
        // row's end - the next row, or past the table the struct's following members - is not used)
        for (int j0 = 0; j0 < n; j0 += 16) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            float wv[16];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const f4u t = *reinterpret_cast<const f4u *>(wr + j0 + 4 * u);
                wv[4 * u] = t.x; wv[4 * u + 1] = t.y; wv[4 * u + 2] = t.z; wv[4 * u + 3] = t.w;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) if (j0 + u < n) { sacc[0] += wv[u] * xa[j0 + u]; sacc[1] += wv[u] * xb[j0 + u]; }
        }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float stab = 0.0f;
        int snr = 0;
        if (lane < npart) {
            const float sa = (0.03f * 0.1f * 0.35f) * hx_mbexp(T.mbexp_lo, T.mbexp_hi, (int) ((1.0f / alpha) * hx_mblog(T.mblog, sacc[c]))) + pc.wabs;
            stab = sa;
            snr = mbe[c] - hx_mblog(T.mblog, pc.wabs + sa);
        }
        int prev = __shfl_up(snr, 1, 64);
        if (lane == 0) prev = 0;
        const bool in = lane < npart;
        int nsnr = hx_wave_sum((in && snr > 0) ? 1 : 0);
        int totsnr = hx_wave_sum(in ? max(-200, snr) : 0);
        int snrvar = hx_wave_sum(in ? abs(snr - prev) : 0);
        int d = 0;
        if (nsnr > 0) {
            int d0 = hx_round(1.3f * (totsnr / npart) - 850);
            int itmp = snrvar / npart;
            int dv = min(500 - itmp, 0);
            d = d0 + dv;
            d = max(d, -2000);
            d = min(d, 600);
        }
        d += 300;
        int dm0 = (300 - d) >> 4;
        int m = lane >> 1;
        int dm = max(dm0 * max(m - 13, 0), 0);
        float a = hx_mbexp(T.mbexp_lo, T.mbexp_hi, d + dm);
        etab_out[64 * c + lane] = (lane < npart2) ? et[c] : 0.0f;
        thr_out[64 * c + lane] = (lane < npart2) ? a * stab : 0.0f;
    }
    FE_WAVE_SYNC();
}

// M/S decision metric before hysteresis: lane = scalefactor band (reference bitallo3.cpp:695-742);
// x0 / x1 = the two channels' lines (LDS)
__device__ __forceinline__ void msmetric_unit(const float *x0, const float *x1, const HxParams *p, const int *t_mblog,
                                              int *out, bool is_short, int sb_start, int sb_n, unsigned run_word, int band_last)
{
    const int lane = threadIdx.x & 63;
    int v = 0;
    if (is_short) {         // short block (reference bitallos.cpp:377-416): lane = (window, sfb)
        const int w = lane >> 4, i = lane & 15;
        int d = 0;
        if (w < 3 && i < p->nsfs) {
            int k = 192 * w + p->startBand_s[i], n = p->nBand_s[i];
            float s0 = 0.0f, s1 = 0.0f;
            for (int j = 0; j < n; j++, k++) {
                float a = x0[k] * x0[k], b = x1[k] * x1[k];
                s0 += (a + b);
                a = fabsf(a - b);
                s1 += a;
            }
            if ((double) s1 > 0.80 * (double) s0) d++;
            if ((double) s1 > 0.95 * (double) s0) d += 2;
        }
        d = hx_wave_sum(d);
        if (lane == 0) *out = (p->nsfs - d) << 10;
        return;
    }
    if (p->alloc1) {        // the first-generation allocator's measure (reference bitallo1.cpp:385-431): bands where one channel dominates count against M/S
        int d = 0;
        if (lane < p->nsf[0]) {
            int k = p->startBand_l[lane], n = p->nBand_l[lane];
            float s0 = 0.0f, s1 = 0.0f;
            for (int j = 0; j < n; j++, k++) {
                float a = x0[k] * x0[k], b = x1[k] * x1[k];
                s0 += (a + b);
                a = fabsf(a - b);
                s1 += a;
            }
            if ((double) s1 > 0.80 * (double) s0) d++;
            if ((double) s1 > 0.95 * (double) s0) d += 2;
        }
        d = hx_wave_sum(d);
        if (lane == 0) *out = p->nsf[0] - 3 * d;
        return;
    }
    // The band's three sums - el = 100 + sum l^2, er = 100 + sum r^2 and the signed t = sum l r, each in line order in the
    // reference - feed four mbLogC arguments only: el + er, max(el, er), es + ed and max(es, ed) (es, ed = (el + er) +- 2 t).
    // So the lanes add the terms of their line runs (the runs of the stream walk's certified band sums, HxParams::lane_run),
    // a segmented scan brings the band's totals to its last lane, and the band lane certifies the four buckets from the
    // intervals the strict sums must lie in (hx_dev.h; the signed sum's half-width comes from sum |l r|); it runs the strict
    // loop only when an interval straddles a bucket boundary (about one band in a hundred).  Before, a wave paid for the widest
    // band's 76 iterations with a third of its lanes in the loop: a third of this kernel's instructions.
    // (tests/cert_sums_check.c checks the certificates on correlated channel pairs of every kind.)
    const bool bandlane = lane < p->nsf[0];
    bool strict = bandlane;
    int mblr = 0, mbsd = 0;
    if (p->ms_flag) {
        const int W = p->run_w;
        const int start = (int) (run_word & 511u) << 1, cnt = (int) ((run_word >> 9) & 7u) << 1, d = (int) (run_word >> 12);
        float sa = 0.0f, sb = 0.0f, sc = 0.0f, sm = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; k += 2) {
            if (k < W) {
                const float2 l = *reinterpret_cast<const float2 *>(x0 + start + k), r = *reinterpret_cast<const float2 *>(x1 + start + k);
                const bool in = k < cnt;
                const float a0 = in ? l.x * l.x : 0.0f, a1 = in ? l.y * l.y : 0.0f, b0 = in ? r.x * r.x : 0.0f, b1 = in ? r.y * r.y : 0.0f;
                const float c0 = in ? l.x * r.x : 0.0f, c1 = in ? l.y * r.y : 0.0f;
                sa += (a0 + a1); sb += (b0 + b1); sc += (c0 + c1); sm += (fabsf(c0) + fabsf(c1));
            }
        }
        const int last4 = 4 * band_last;
        const float SA = hx_lane_read(last4, hx_seg_scan(sa, d, lane)), SB = hx_lane_read(last4, hx_seg_scan(sb, d, lane));
        const float SC = hx_lane_read(last4, hx_seg_scan(sc, d, lane)), SM = hx_lane_read(last4, hx_seg_scan(sm, d, lane));
        if (bandlane) {
            const float du = hx_cert_delta(sb_n + 1, W);      // (the 100 in front is one more term and one more addition)
            const float tel = 100.0f + SA, ter = 100.0f + SB;
            const float e1 = tel * du, e2 = ter * du, e3 = SM * du;
            // (a sum of non-negative terms that starts at 100 never falls below 100: rounding is monotone)
            const float el_lo = fmaxf(tel - e1, 100.0f), el_hi = tel + e1, er_lo = fmaxf(ter - e2, 100.0f), er_hi = ter + e2;
            const float t_lo = SC - e3, t_hi = SC + e3;
            const float tl2 = t_lo + t_lo, th2 = t_hi + t_hi;
            const float p1_lo = el_lo + er_lo, p1_hi = el_hi + er_hi;
            const float p2_lo = fmaxf(el_lo, er_lo), p2_hi = fmaxf(el_hi, er_hi);
            const float es_lo = p1_lo + tl2, es_hi = p1_hi + th2, ed_lo = p1_lo - th2, ed_hi = p1_hi - tl2;
            const float p3_lo = es_lo + ed_lo, p3_hi = es_hi + ed_hi;
            // (one of es, ed is el + er plus something non-negative, rounded: max(es, ed) >= el + er)
            const float p4_lo = fmaxf(fmaxf(es_lo, ed_lo), p1_lo), p4_hi = fmaxf(es_hi, ed_hi);
            const bool ok = p3_lo > 0.0f && p4_lo > 0.0f && (hx_f2bits(p1_lo) >> 15) == (hx_f2bits(p1_hi) >> 15) && (hx_f2bits(p2_lo) >> 15) == (hx_f2bits(p2_hi) >> 15)
                            && (hx_f2bits(p3_lo) >> 15) == (hx_f2bits(p3_hi) >> 15) && (hx_f2bits(p4_lo) >> 15) == (hx_f2bits(p4_hi) >> 15);
            if (ok) {       // every point of a certified interval has the strict value's log: take the lower ends
                strict = false;
                mblr = hx_mblog(t_mblog, p1_lo) - hx_mblog(t_mblog, p2_lo);
                mbsd = hx_mblog(t_mblog, p3_lo) - hx_mblog(t_mblog, p4_lo);
            }
        }
    }
    if (strict) {
        int k = sb_start;
        float el = 100.0f, er = 100.0f, t = 0.0f;
        for (int j = 0; j < sb_n; j++, k++) {
            float a = x0[k] * x0[k], b = x1[k] * x1[k], c = x0[k] * x1[k];
            el += a; er += b; t += c;
        }
        float es, ed;
        es = ed = el + er;
        t = t + t;
        es = es + t;
        ed = ed - t;
        mblr = hx_mblog(t_mblog, el + er) - hx_mblog(t_mblog, el > er ? el : er);
        mbsd = hx_mblog(t_mblog, es + ed) - hx_mblog(t_mblog, es > ed ? es : ed);
    }
    if (bandlane) {
        int psd = max(75 - abs(mblr - 120), 0);
        mbsd = min(mbsd, (mbsd >> 1) + 120);
        mbsd += psd;
        v = sb_n * (mblr - mbsd);
    }
    v = hx_wave_sum(v);
    if (lane == 0) *out = v;
}

// DIRECT = false: the granule's subband samples are staged in LDS by 16-byte loads and picked up from there (24 KB of LDS per
// workgroup); DIRECT = true: every lane loads its own 2 x 18 samples from global memory (8-byte loads; 14.8 KB).  Measured
// (EXPERIMENTS.md, round 4): alone on the chip the direct form is 11 % faster (configs 3 - 5, where the stream walk's
// low-footprint kernel leaves the front end no room beside it: +3.6 / +2.7 % per step); beside the resident stream walk of
// config 2 it is 0.7 % slower per step - more of its smaller workgroups fit next to the walk's waves and take issue slots
// from them.  hx_cabi.hip launches the form that goes with the stream-walk kernel it chose.
template <bool DIRECT>
__device__ __forceinline__ void spec_granule(const float *__restrict__ sb, const HxStream *__restrict__ st,
                                             const HxParams *__restrict__ prm, const HxGlobalTabs *__restrict__ gt,
                                             const unsigned char *__restrict__ bt,
                                             float *__restrict__ xr, float *__restrict__ etab_out, float *__restrict__ thr_out,
                                             int *__restrict__ msbase, int NG, int SG)
{
    // in: [ch][S[g-3] | S[g-2]][576] subband samples; the first 2 x 576 floats are reused as the
    // spectrum [ch][576] once every lane holds its inputs in registers (DIRECT: the spectrum only)
    // two granules per workgroup, one wavefront each, independent of each other but for the lookup tables they share
    __shared__ __attribute__((aligned(16))) float in_s[2][DIRECT ? 1 : 2][2][576];
    __shared__ float xtab_s[2][2][64];
    __shared__ float es_s[2][3][64];
    __shared__ SpecTabs T;
    // (the wave index is wave-uniform, which the compiler cannot see in threadIdx: through readfirstlane the granule's
    // addresses are scalar arithmetic)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sbnd = lane & 31, ch = lane >> 5;
    for (int i = threadIdx.x; i < 256; i += 128) { T.mblog[i] = gt->mblog[i]; T.mbexp_lo[i] = gt->mbexp_lo[i]; T.mbexp_hi[i] = gt->mbexp_hi[i]; }
    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own; neighbouring granules share a subband
    // block, so every XCD gets a contiguous piece of the grid: workgroups b and b + 8, launched together on one XCD, are
    // neighbours in the stream, and the second read of the shared block hits that L2 (speed only: any placement is correct)
    const unsigned nwg = gridDim.x, cpx = nwg >> 3;
    const unsigned wg = (blockIdx.x < (cpx << 3)) ? (blockIdx.x & 7) * cpx + (blockIdx.x >> 3) : blockIdx.x;
    const long long sg = (long long) wg * 2 + wv;                   // (s, g); S * NG is even
    const int g = (int) (sg % NG), s = (int) (sg / NG);
    const HxParams *p = prm + __builtin_amdgcn_readfirstlane(st[s].cls);      // wave-uniform: table reads become scalar loads
    float (*in)[2][576] = in_s[wv];
    // this lane's entries of the psy and band tables: requested now, needed after the transform
    PsyLane pc;
    pc.i0 = p->psyL.pstart[lane]; pc.nsum = p->psyL.nsum[lane]; pc.off = p->psyL.off[lane]; pc.cnt = p->psyL.cnt[lane];
    pc.row = p->psyL.row[lane]; pc.wabs = p->psyL.w[lane];
    const int sb_start = p->startBand_l[min(lane, 22)], sb_n = p->nBand_l[min(lane, 21)];
    const unsigned run_word = p->lane_run[lane];            // the lane's line run and its band's last lane, for the stereo metric's sums
    const int band_last = p->band_last_lane[min(lane, 21)];
    __syncthreads();        // the tables (the only workgroup barrier: from here on each wave is on its own)
    const int nsb = p->nsb_ms0;
    const int btype = bt[sg];
    float g1[DIRECT ? 18 : 1], g2[DIRECT ? 18 : 1];
    const float *x1, *x2;
    if constexpr (DIRECT) {
        // every lane takes its subband's 18 + 18 samples straight from global memory (72 contiguous bytes per block, 8-byte
        // aligned; the wave's 64 lanes cover two contiguous 2304-byte runs per channel)
#ifdef HX_MOCK_NOSB_LOAD
        // (timing experiment: every workgroup reads the same two blocks - cache hits, no HBM traffic for the subband samples)
        const float2 *b1 = reinterpret_cast<const float2 *>(sb + (long long) ch * SG * 576 + sbnd * 18);
#else
        const float2 *b1 = reinterpret_cast<const float2 *>(sb + ((long long) (s * 2 + ch) * SG + g) * 576 + sbnd * 18);
#endif
        const float2 *b2 = b1 + 288;
#pragma unroll
        for (int i = 0; i < 9; i++) { const float2 a = b1[i], b = b2[i]; g1[2 * i] = a.x; g1[2 * i + 1] = a.y; g2[2 * i] = b.x; g2[2 * i + 1] = b.y; }
        x1 = g1; x2 = g2;
    } else {
        {   // 2 x 1152 contiguous floats per channel, 16 bytes per lane and load
            float4 v[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int e = lane + 64 * k, c = e / 288, r = e - 288 * c;       // 288 float4 per channel
                v[k] = reinterpret_cast<const float4 *>(sb + ((long long) (s * 2 + c) * SG + g) * 576)[r];
            }
#pragma unroll
            for (int k = 0; k < 9; k++) reinterpret_cast<float4 *>(&in[0][0][0])[lane + 64 * k] = v[k];
        }
        FE_WAVE_SYNC();
        x1 = &in[ch][0][sbnd * 18];                    // S[g-3]
        x2 = &in[ch][1][sbnd * 18];                    // S[g-2]
    }
    float y[18], f[18];
    const bool act = sbnd < nsb;
    {
        float p1[18], p2[18];
        const bool inv = (sbnd & 1) != 0;       // odd subbands: negate odd time slots
#pragma unroll
        for (int i = 0; i < 18; i++) {
            float a = x1[i], b = x2[i];
            if (inv && (i & 1)) { a = -a; b = -b; }
            p1[i] = a; p2[i] = b;
        }
        FE_WAVE_SYNC();                         // everyone has its inputs: `in` may be overwritten
        if (!act) {
#pragma unroll
            for (int i = 0; i < 18; i++) y[i] = 0.0f;
        } else if (btype != 2) {
            const float *w = p->win[btype];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                f[j] = w[26 - j] * p2[8 - j] + w[27 + j] * p2[9 + j];
                f[9 + j] = w[j] * p1[j] + w[17 - j] * p1[17 - j];
            }
            mdct_kernel<Cos9>(p, p->mdct_pre18, p->mdct_odd18, f, y);
        } else {        // short: three overlapping 12-tap windows (reference hwin.c:228-278)
            const float *w = p->win[2];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                f[q] = w[8 - q] * p1[14 - q] + w[9 + q] * p1[15 + q];
                f[3 + q] = w[q] * p1[6 + q] + w[5 - q] * p1[11 - q];
                f[6 + q] = w[8 - q] * p2[2 - q] + w[9 + q] * p2[3 + q];
                f[9 + q] = w[q] * p1[12 + q] + w[5 - q] * p1[17 - q];
                f[12 + q] = w[8 - q] * p2[8 - q] + w[9 + q] * p2[9 + q];
                f[15 + q] = w[q] * p2[q] + w[5 - q] * p2[5 - q];
            }
#pragma unroll
            for (int w = 0; w < 3; w++) mdct_kernel<Cos3>(p, p->mdct_pre6, p->mdct_odd6, f + 6 * w, y + 6 * w);
        }
    }
    // alias reduction between subband k (lane) and k+1: x[17-i] with next lane's x[i]
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float up = __shfl_down(y[i], 1, 64);            // next subband's element i
        float dn = __shfl_up(y[17 - i], 1, 64);         // previous subband's element 17-i
        float cs = p->csa[0][i], ca = p->csa[1][i];
        float a = y[17 - i], b = y[i];
        float na = a, nb = b;
        if (btype != 2) {                                   // no alias reduction on short blocks
            if (sbnd < nsb - 1) na = a * cs + up * ca;      // upper edge of this band
            else if (sbnd == nsb - 1) na = a * cs;          // last coded band: half butterfly
            if (sbnd >= 1 && sbnd < nsb) nb = b * cs - dn * ca;     // lower edge (pairs with band-1)
        }
        y[17 - i] = na;
        y[i] = nb;
    }
    float *xl = &in[0][0][0];                   // spectrum [ch][576]
    if (btype != 2) {
        float *o = xl + ch * 576 + sbnd * 18;
#pragma unroll
        for (int i = 0; i < 18; i++) o[i] = y[i];
    } else {                                    // [3 windows][192], line = 6*sb + k
        float *o = xl + ch * 576 + sbnd * 6;
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
            for (int k = 0; k < 6; k++) o[192 * w + k] = y[6 * w + k];
    }
    FE_WAVE_SYNC();
    {   // spectrum to global memory, 16 bytes per lane and store
        float4 *dst = reinterpret_cast<float4 *>(xr + sg * 1152);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int e = lane + 64 * k;
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (e < 288) __builtin_nontemporal_store(reinterpret_cast<const f4v *>(xl)[e], reinterpret_cast<f4v *>(dst) + e);     // read once, a kernel later
        }
    }
    if (btype != 2) psy_long2(xl, p, T, pc, etab_out + sg * 128, thr_out + sg * 128, xtab_s[wv]);
    else {
        const HX_LDS float *xs = (const HX_LDS float *) xl;
        HX_LDS float (*ess)[64] = (HX_LDS float (*)[64]) es_s[wv];
        const HX_GLB HxParams *pg = (const HX_GLB HxParams *) p;
        psy_short(xs, pg, (HX_GLB float *) (etab_out + sg * 128), (HX_GLB float *) (thr_out + sg * 128), ess);
        psy_short(xs + 576, pg, (HX_GLB float *) (etab_out + sg * 128 + 64), (HX_GLB float *) (thr_out + sg * 128 + 64), ess);
    }
    msmetric_unit(xl, xl + 576, p, T.mblog, msbase + sg, btype == 2, sb_start, sb_n, run_word, band_last);
}

#define HX_K4(name, direct) \
__global__ __launch_bounds__(128) void name(const float *__restrict__ sb, const HxStream *__restrict__ st, const HxParams *__restrict__ prm, \
                                           const HxGlobalTabs *__restrict__ gt, const unsigned char *__restrict__ bt, float *__restrict__ xr, \
                                           float *__restrict__ etab_out, float *__restrict__ thr_out, int *__restrict__ msbase, int NG, int SG) \
{ spec_granule<direct>(sb, st, prm, gt, bt, xr, etab_out, thr_out, msbase, NG, SG); }
HX_K4(k_spec, false)
HX_K4(k_spec_direct, true)

// K5a: the frame's stereo decision (joint-stereo streams), serial per stream over its granules, and the hand-over
// of the pre-echo memory between calls.  The L/R-vs-M/S metric of a granule gets a +-5000 hysteresis from the
// previous long granule; a short granule takes none and clears it (reference bitallo3.cpp:693-698,743-751); an
// MPEG-1 frame is coded M/S when its two granules' values sum to >= 0 (mp3enc.cpp:1538-1546), an MPEG-2 frame by
// its one granule.  Depends on front-end data only, so it runs here and not in the per-stream allocator walk.
// (Round 6: the kernel also rolls the stream's carries - the last three granules of subband samples to slots 0..2, the last 480
// input samples into the stream state - which was k_carry's launch: k_spec, the last reader of the subband buffer, is through
// when this kernel starts, and 63 of its 64 lanes had nothing to do.)
__global__ __launch_bounds__(64) void k_msscan(HxStream *__restrict__ st, const HxParams *__restrict__ prm, const int *__restrict__ msbase,
                                               const unsigned char *__restrict__ bt, unsigned char *__restrict__ msflag, int *__restrict__ msdec,
                                               const float *__restrict__ thr, float *__restrict__ thrprev, int NG, int lsf,
                                               float *__restrict__ sb, int SG, const int16_t *__restrict__ pcm, long long nsamp,
                                               const float *__restrict__ pcmf, int nchan)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    HxStream *ss = st + s;
    const long long g0 = (long long) s * NG;
    if (lane == 0) {
        const int on = prm[ss->cls].ms_flag, plain = prm[ss->cls].alloc1;    // (the first-generation allocator's measure takes no hysteresis)
        int mem = ss->ms_memory;
        auto frame = [&](int b1, int b2, int v1, int v2, int *m1o, int *m2o) {      // one pair of granules; returns the two flags
            int m1 = 0, m2 = 0;
            if (on) {
                m1 = v1;
                if (plain) { }
                else if (b1 == 2) mem = 0; else { m1 += mem; mem = (m1 > 0) ? 5000 : -5000; }
                m2 = v2;
                if (plain) { }
                else if (b2 == 2) mem = 0; else { m2 += mem; mem = (m2 > 0) ? 5000 : -5000; }
            }
            *m1o = m1; *m2o = m2;
            const unsigned f1 = (on && (lsf ? m1 : m1 + m2) >= 0), f2 = (on && (lsf ? m2 : m1 + m2) >= 0);
            return f1 | (f2 << 8);
        };
        int g = 0;
        // eight granules per round of loads and stores (the rows are 16-byte aligned when NG % 8 == 0)
        if ((NG & 7) == 0)
            for (; g + 8 <= NG; g += 8) {
                const int4 va = *reinterpret_cast<const int4 *>(msbase + g0 + g), vb = *reinterpret_cast<const int4 *>(msbase + g0 + g + 4);
                const uint2 bb = *reinterpret_cast<const uint2 *>(bt + g0 + g);
                int4 da, db;
                const unsigned fa = frame(bb.x & 255, (bb.x >> 8) & 255, va.x, va.y, &da.x, &da.y);
                const unsigned fb = frame((bb.x >> 16) & 255, bb.x >> 24, va.z, va.w, &da.z, &da.w);
                const unsigned fc = frame(bb.y & 255, (bb.y >> 8) & 255, vb.x, vb.y, &db.x, &db.y);
                const unsigned fd = frame((bb.y >> 16) & 255, bb.y >> 24, vb.z, vb.w, &db.z, &db.w);
                *reinterpret_cast<int4 *>(msdec + g0 + g) = da;
                *reinterpret_cast<int4 *>(msdec + g0 + g + 4) = db;
                *reinterpret_cast<uint2 *>(msflag + g0 + g) = make_uint2(fa | (fb << 16), fc | (fd << 16));
            }
        for (; g < NG; g += 2) {
            int m1, m2;
            const unsigned f = frame(bt[g0 + g], bt[g0 + g + 1], msbase[g0 + g], msbase[g0 + g + 1], &m1, &m2);
            msdec[g0 + g] = m1; msdec[g0 + g + 1] = m2;
            msflag[g0 + g] = (unsigned char) (f & 1); msflag[g0 + g + 1] = (unsigned char) (f >> 8);
        }
        ss->ms_memory = mem;
    }
    // pre-echo memory ("ecsave", reference spdsmr.c:112-117,283-298): this call's first granule is clamped against
    // the stream's carried values, which then become the doubled unclamped thresholds of this call's last granule
    // (long), or its doubled window-2 sums in entries 0..11 (short)
    const float *last = thr + (g0 + NG - 1) * 128;
    const int lastbt = bt[g0 + NG - 1];
    for (int i = lane; i < 128; i += 64) {
        const float old = (&ss->thr_prev[0][0])[i];
        thrprev[(long long) s * 128 + i] = old;
        float nw = old;
        if (lastbt != 2) nw = 2.0f * last[i];
        else if ((i & 63) < 12) nw = 2.0f * last[(i & 64) + 24 + (i & 63)];
        (&ss->thr_prev[0][0])[i] = nw;
    }
    // the carries into the next call
    for (int ch = 0; ch < 2; ch++) {
        float *base = sb + (long long) (s * 2 + ch) * SG * 576;
        for (int e = lane; e < 576; e += 64)
            for (int k = 0; k < 3; k++) base[k * 576 + e] = base[(NG + k) * 576 + e];
        const int16_t *src = pcm + (long long) s * nsamp * nchan + ch;
        for (int i = lane; i < 480 && ch < nchan; i += 64) {
            const long long n = nsamp - 480 + i;
            // fewer than 480 new samples never happens (a frame is 1152), so all come from this batch
            ss->pcm_hist[ch][i] = pcmf ? pcmf[((long long) s * nsamp + n) * nchan + ch] : (float) src[nchan * n];
        }
    }
    if (lane == 0) ss->frames_in += (int) (nsamp / 1152);
}

// K5b: everything the allocator does at the start of a long-block granule that does not depend on its carried
// state, one wavefront per (stream, granule): magnitudes and signs of the lines in the representation the frame
// is coded in (L / R, or M = L + R, S = L - R: reference l3math.c:449-470,905-930), band energies in line order
// (bitallo3.cpp:816-864,902-985), x^(3/4) of every line with the band maxima and the zero-gain steps
// (:878-896, pow34.c:132-186), and the masking thresholds after pre-echo control (spdsmr.c:275-318).  The
// short-block granules are skipped (their allocator starts from the raw spectrum).
#define PREP_GPB 4      // granules (wavefronts) per workgroup: they share one copy of the lookup tables
__global__ __launch_bounds__(64 * PREP_GPB) void k_prep(const float *__restrict__ xr, float *__restrict__ xmag_dbg, float *__restrict__ x34o, unsigned *__restrict__ sgn,
                                             HxBandPrep *__restrict__ band, const HxStream *__restrict__ st,
                                             const HxParams *__restrict__ prm, const HxGlobalTabs *__restrict__ gt,
                                             const unsigned char *__restrict__ bt, const unsigned char *__restrict__ msflag,
                                             const float *__restrict__ etab, const float *__restrict__ thr,
                                             const float *__restrict__ thrprev, int NG, long long nunits)
{
    // Per wave only the squares that the band lanes add up live in LDS (one pair of channels at a time: L / R, then
    // M / S); magnitudes, x^(3/4) and signs stay in the registers of the lane that owns the lines, from the load
    // to the store.  The gather tables are staged once per workgroup.
    __shared__ __attribute__((aligned(16))) float sq[PREP_GPB][2][576];
    __shared__ int xmax[PREP_GPB][2][22];
    __shared__ float t_exp[256], t_a[16], t_b[16];
    __shared__ int t_mblog[256];
    for (int i = threadIdx.x; i < 256; i += 64 * PREP_GPB) { t_exp[i] = gt->pow34_exp[i]; t_mblog[i] = gt->mblog[i]; }
    if (threadIdx.x < 16) { t_a[threadIdx.x] = gt->pow34_a[threadIdx.x]; t_b[threadIdx.x] = gt->pow34_b[threadIdx.x]; }
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (lane < 44) xmax[wv][lane / 22][lane % 22] = 0;
    __syncthreads();
    const long long unit = (long long) blockIdx.x * PREP_GPB + wv;      // (s, g)
    if (unit >= nunits) return;
    const int g = (int) (unit % NG), s = (int) (unit / NG);
    const int btype = bt[unit];
    if (btype == 2) return;
    // (from here on the wave works alone: LDS hand-overs inside a wave need no workgroup barrier)
#define WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
    const HxParams *p = prm + __builtin_amdgcn_readfirstlane(st[s].cls);
    if (p->alloc1) return;      // the first-generation allocator starts from the raw spectrum
    const int ms = msflag[unit];
    const int two = p->nchan == 2;
    const int nsf0 = p->nsf[0];
    // lines that get magnitudes / x^(3/4), bands that get energies / maxima (reference: nbmax, nbmax2 / nbmax3 ...)
    const int nl_mag0 = ms ? (p->hf_flag ? p->startBand_l[22] : p->nbmax[0]) : p->nbmax3[0];
    const int nl_mag1 = ms ? nl_mag0 : (two ? p->nbmax3[1] : 0);
    const int nl_p0 = ms ? p->nbmax2[0] : p->nbmax3[0], nl_p1 = two ? (ms ? p->nbmax2[1] : p->nbmax3[1]) : 0;
    const int nb_e0 = ms ? nsf0 : p->nsf3[0], nb_e1 = ms ? nsf0 : (two ? p->nsf3[1] : 0);
    const int nb_z0 = ms ? p->nsf2[0] : p->nsf3[0], nb_z1 = two ? (ms ? p->nsf2[1] : p->nsf3[1]) : 0;
    const float *x = xr + unit * 1152;
    float (*sqw)[576] = sq[wv];
    // lane l owns lines 4 (l + 64 k) .. + 3 of both channels, k = 0..2 (144 groups of four per channel)
    float a0[3][4], a1[3][4];           // magnitudes in the coded representation
    unsigned s0[3], s1[3];              // sign bytes
    {
        float4 lv[3], rv[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = min(lane + 64 * k, 143);
            lv[k] = reinterpret_cast<const float4 *>(x)[e];
            rv[k] = reinterpret_cast<const float4 *>(x + 576)[e];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = lane + 64 * k;
            const float l4[4] = {lv[k].x, lv[k].y, lv[k].z, lv[k].w}, r4[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
            float t0[4], t1[4];
            s0[k] = s1[k] = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = 4 * e + c;
                const float l = l4[c], r = r4[c];
                a0[k][c] = l; a1[k][c] = r; t0[c] = t1[c] = 0.0f;
                if (ms) {
                    if (j < nl_mag0) {
                        t0[c] = l * l;
                        t1[c] = r * r;
                        float m = (l + r), d = (l - r);
                        if (m < 0.0f) { s0[k] |= 1u << (8 * c); m = -m; }
                        if (d < 0.0f) { s1[k] |= 1u << (8 * c); d = -d; }
                        a0[k][c] = m; a1[k][c] = d;
                    }
                } else {
                    if (j < nl_mag0) { float v = l; if (!(v >= 0.0f)) { s0[k] |= 1u << (8 * c); v = -v; } a0[k][c] = v; t0[c] = v * v; }
                    if (j < nl_mag1) { float v = r; if (!(v >= 0.0f)) { s1[k] |= 1u << (8 * c); v = -v; } a1[k][c] = v; t1[c] = v * v; }
                }
            }
            if (e < 144) {
                reinterpret_cast<float4 *>(sqw[0])[e] = make_float4(t0[0], t0[1], t0[2], t0[3]);
                reinterpret_cast<float4 *>(sqw[1])[e] = make_float4(t1[0], t1[1], t1[2], t1[3]);
            }
        }
    }
    WAVE_SYNC();
    // band energies: lane (ch, sfb) adds its band's squares in line order - L / R first, then (joint stereo) M / S
    const int ch = lane >> 5, i = lane & 31;
    const int cbw = (i < 22) ? p->look_log_cbwmb[i] : 0;
    const bool eband = i < (ch ? nb_e1 : nb_e0);
    const int b0 = eband ? p->startBand_l[i] : 0, bn = eband ? p->nBand_l[i] : 0;
    float e_lr = 0.0f;
    int n0 = 0, n0ms = 0;
    if (eband) {
        e_lr = band_sum(&sqw[ch][b0], bn, 0.0f);
        n0 = hx_mblog(t_mblog, e_lr) - cbw;
    }
    if (ms) {
        WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = lane + 64 * k;
            if (e < 144) {
                // (lines past nl_mag0 were left raw above; their squares are never summed)
                reinterpret_cast<float4 *>(sqw[0])[e] = make_float4(a0[k][0] * a0[k][0], a0[k][1] * a0[k][1], a0[k][2] * a0[k][2], a0[k][3] * a0[k][3]);
                reinterpret_cast<float4 *>(sqw[1])[e] = make_float4(a1[k][0] * a1[k][0], a1[k][1] * a1[k][1], a1[k][2] * a1[k][2], a1[k][3] * a1[k][3]);
            }
        }
        WAVE_SYNC();
        if (eband) n0ms = hx_mblog(t_mblog, band_sum(&sqw[ch][b0], bn, 0.0f)) - cbw;
    }
    // x^(3/4) of the coded magnitudes and the band maxima (bit patterns of non-negative floats order like integers)
    float q0[3][4], q1[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = lane + 64 * k;
        unsigned bl = 0;
        if (e < 144) bl = reinterpret_cast<const unsigned *>(p->band_of_line)[e];       // the four lines' bands
        // The band maximum of x^(3/4) is the x^(3/4) of the band's largest magnitude: the piecewise-linear fit is monotone
        // over every non-negative float (checked exhaustively: tools/check_pow34_monotone.cpp).  So the lines contribute
        // their magnitudes (sign bit off: lines past the magnitude range are raw, and the fit ignores the sign), and the
        // band lane evaluates the fit once - 44 evaluations per granule instead of 1152.  The lines' own x^(3/4) is only
        // formed for the tests' tap (the allocator's helper wave computes it for itself).
        // A lane's four lines are two pairs, and a pair never straddles a band or the end of a coded range (bands start on even
        // lines and have even widths): one LDS atomic per pair - per four lines when both pairs are in one band - instead of one
        // per line.  (Atomics of a wave on one address are served one lane after the other: in a wide band that was up to 40
        // passes per instruction, 24 instructions per lane.)
        {
            const int b0 = bl & 255, b2 = (bl >> 16) & 255, j0 = 4 * e, j2 = 4 * e + 2;
            int m00 = max(__float_as_int(a0[k][0]) & 0x7FFFFFFF, __float_as_int(a0[k][1]) & 0x7FFFFFFF);
            const int m02 = max(__float_as_int(a0[k][2]) & 0x7FFFFFFF, __float_as_int(a0[k][3]) & 0x7FFFFFFF);
            int m10 = max(__float_as_int(a1[k][0]) & 0x7FFFFFFF, __float_as_int(a1[k][1]) & 0x7FFFFFFF);
            const int m12 = max(__float_as_int(a1[k][2]) & 0x7FFFFFFF, __float_as_int(a1[k][3]) & 0x7FFFFFFF);
            // (inside / outside the coded ranges: part of what makes a group - with -HF the range ends inside the last band's run
            // of the line-to-band table, which maps everything above band 20 to band 21)
            const int in0 = (j0 < nl_p0) | ((j0 < nl_p1) << 1), in2 = (j2 < nl_p0) | ((j2 < nl_p1) << 1);
            const bool one = b0 == b2 && in0 == in2;
            if (one) { m00 = max(m00, m02); m10 = max(m10, m12); }
            // ... and per pair or four of neighbouring lanes whose lines lie in one band and on one side of the coded ranges' ends
            // (they are then also all below line 576 or all above): the group's first lane brings the maximum of the group
            bool issue = true;
            {
                const int key = one ? (b0 | (in0 << 8)) : -1 - lane;       // (a lane whose lines straddle two bands joins no group)
#define PREP_QP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
                const bool pair = PREP_QP(key, 0xB1) == key;                            // quad_perm [1,0,3,2]: the lane beside this one
                if (pair) { m00 = max(m00, PREP_QP(m00, 0xB1)); m10 = max(m10, PREP_QP(m10, 0xB1)); }
                const bool quad = pair && PREP_QP((int) pair, 0x4E) != 0 && PREP_QP(key, 0x4E) == key;      // quad_perm [2,3,0,1]: the other pair
                if (quad) { m00 = max(m00, PREP_QP(m00, 0x4E)); m10 = max(m10, PREP_QP(m10, 0x4E)); }
#undef PREP_QP
                issue = quad ? (lane & 3) == 0 : (pair ? (lane & 1) == 0 : true);
            }
            if (issue && e < 144 && j0 < nl_p0) atomicMax(&xmax[wv][0][b0], m00);
            if (issue && e < 144 && j0 < nl_p1) atomicMax(&xmax[wv][1][b0], m10);
            if (!one) {
                if (e < 144 && j2 < nl_p0) atomicMax(&xmax[wv][0][b2], m02);
                if (e < 144 && j2 < nl_p1) atomicMax(&xmax[wv][1][b2], m12);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = 4 * e + c;
            q0[k][c] = q1[k][c] = 0.0f;
            if (x34o) {
                if (e < 144 && j < nl_p0) q0[k][c] = hx_pow34(t_a, t_b, t_exp, a0[k][c]);
                if (e < 144 && j < nl_p1) q1[k][c] = hx_pow34(t_a, t_b, t_exp, a1[k][c]);
            }
        }
    }
    WAVE_SYNC();
    int gz = 0;
    float xm = 0.0f;
    if (i < 22) xm = hx_pow34(t_a, t_b, t_exp, __int_as_float(xmax[wv][ch][i]));
    if (i < (ch ? nb_z1 : nb_z0)) gz = max(0, hx_round((0.017716950f * hx_mblog(t_mblog, xm) + (104.585000f - 100.0f + 8.0f))));
    // masking threshold of the band: the two partitions' thresholds, each clamped against twice the previous
    // granule's unless this is a stop block, weighted by the partitions' energies
    int mmb = 0;
    if (i < 21) {
        const float2 th = reinterpret_cast<const float2 *>(thr + unit * 128 + ch * 64)[i];
        const float2 en = reinterpret_cast<const float2 *>(etab + unit * 128 + ch * 64)[i];
        const float2 pv = reinterpret_cast<const float2 *>((g == 0 ? thrprev + (long long) s * 128 : thr + (unit - 1) * 128) + ch * 64)[i];
        float s1v = th.x, s2v = th.y;
        const float t1 = (g == 0) ? pv.x : 2.0f * pv.x, t2 = (g == 0) ? pv.y : 2.0f * pv.y;
        if (btype != 3) {
            if (s1v > t1) { const float f = 0.1f * s1v; s1v = t1; if (s1v < f) s1v = f; }
            if (s2v > t2) { const float f = 0.1f * s2v; s2v = t2; if (s2v < f) s2v = f; }
        }
        float emax = en.x;
        if (emax < en.y) emax = en.y;
        mmb = hx_mblog(t_mblog, (en.x * s1v + en.y * s2v) / emax);
    }
    HxBandPrep *bp = band + unit;
    if (i < 22) {
        bp->xsxx[ch][i] = e_lr; bp->x34max[ch][i] = xm; bp->n0[ch][i] = n0; bp->n0ms[ch][i] = n0ms;
        bp->gzero[ch][i] = gz; bp->maskmb[ch][i] = mmb;
    }
    {   // The signs to their buffer, straight from the owning lanes.  Magnitudes and x^(3/4) are not stored: the allocator's
        // helper wave forms them again from the spectrum it fetches, in time it would otherwise spend waiting - cheaper
        // than 4.8 GB of stores and as many loads per launch.  (xmag_dbg, x34o: the tests' taps.)
        float4 *dq = reinterpret_cast<float4 *>(x34o + unit * 1152);     // (x34o: the tests' tap; the allocator's helper wave computes x^(3/4) again)
        unsigned *ds = sgn + unit * (2 * HX_SGN_WORDS);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = lane + 64 * k;
            if (e < 144) {
                if (xmag_dbg) {
                    float4 *dx = reinterpret_cast<float4 *>(xmag_dbg + unit * 1152);
                    dx[e] = make_float4(a0[k][0], a0[k][1], a0[k][2], a0[k][3]);
                    dx[144 + e] = make_float4(a1[k][0], a1[k][1], a1[k][2], a1[k][3]);
                }
                if (x34o) {
                    dq[e] = make_float4(q0[k][0], q0[k][1], q0[k][2], q0[k][3]);
                    dq[144 + e] = make_float4(q1[k][0], q1[k][1], q1[k][2], q1[k][3]);
                }
            }
            // signs as one bit per line, line order: a lane's four lines are a nibble (its sign bytes' low bits), eight
            // neighbouring lanes a word - OR over the group of eight on the DPP path, the group's first lane stores it
            unsigned w0 = ((s0[k] & 1u) | ((s0[k] >> 7) & 2u) | ((s0[k] >> 14) & 4u) | ((s0[k] >> 21) & 8u)) << (4 * (lane & 7));
            unsigned w1 = ((s1[k] & 1u) | ((s1[k] >> 7) & 2u) | ((s1[k] >> 14) & 4u) | ((s1[k] >> 21) & 8u)) << (4 * (lane & 7));
            w0 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w0, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
            w1 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w1, 0xB1, 0xf, 0xf, true);
            w0 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w0, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
            w1 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w1, 0x4E, 0xf, 0xf, true);
            w0 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w0, 0x141, 0xf, 0xf, true);     // row_half_mirror: the other quad of the eight
            w1 |= (unsigned) __builtin_amdgcn_update_dpp(0, (int) w1, 0x141, 0xf, 0xf, true);
            if ((lane & 7) == 0 && e < 144) { ds[e >> 3] = w0; ds[HX_SGN_WORDS + (e >> 3)] = w1; }
        }
    }
#undef WAVE_SYNC
}

#endif      // HX_FRONT_PART & 2
#if HX_FRONT_PART & 1
// K0 (only when some stream asked for it, E_CONTROL filter_select = 1): the input DC blocker
// y = x - d, d += alpha * y (reference filter2.c:116-121,137-144).  A first-order recurrence
// evaluated in the reference's order, so it is sequential per channel: one lane per
// (stream, channel) walks its samples; streams without the filter are converted to float only.
__global__ void k_dcfilter(const int16_t *__restrict__ pcm, const float *__restrict__ pcm32, long long nsamp,
                           HxStream *__restrict__ st, const HxParams *__restrict__ prm, float *__restrict__ pcmf, int S, int nchan)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nchan * S) return;
    const int s = u / nchan, ch = u - s * nchan;
    HxStream *ss = st + s;
    const HxParams *p = prm + ss->cls;
    const int16_t *src = pcm + (long long) s * nsamp * nchan + ch;
    const float *srcf = pcm32 + (long long) s * nsamp * nchan + ch;
    float *dst = pcmf + (long long) s * nsamp * nchan + ch;
    if (!p->filter_dc) {
        for (long long n = 0; n < nsamp; n++) dst[nchan * n] = pcm32 ? srcf[nchan * n] : (float) src[nchan * n];
        return;
    }
    const float alpha = p->filter_alpha;
    float d = ss->dc[ch];
    for (long long n0 = 0; n0 < nsamp; n0 += 8) {      // nsamp is a multiple of 1152
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = pcm32 ? srcf[nchan * (n0 + k)] : (float) src[nchan * (n0 + k)];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float t = x[k] - d;
            d = d + alpha * t;
            dst[nchan * (n0 + k)] = t;
        }
    }
    ss->dc[ch] = d;
}

// Gate of a pipelined submit (hx_batch_submit_*): holds the stream it is launched on until `need` workgroups of
// the previous call's allocator kernel have started, i.e. until that kernel occupies its share of the chip.
// The front-end kernels behind the gate then queue for the slots that finishing streams free, and run in the
// allocator kernel's tail; released earlier they would take LDS away from allocator workgroups that have not
// started yet.  The counter runs over all launches of the batch and may wrap: `base` is its value when the
// previous launch began, and the distance is compared as unsigned.  Gives up after ~50 ms (late is harmless, a
// hang is not) and counts that in *timeouts, so the caller can see that the overlap degraded.
__global__ void k_gate(const unsigned *started_counter, unsigned base, unsigned need, int *timeouts)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();        // 100 MHz
    while (__hip_atomic_load(started_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base < need) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 5000000LL) { atomicAdd(timeouts, 1); break; }
    }
}
#endif      // HX_FRONT_PART & 1
