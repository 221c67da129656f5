// hx_alloc.hip - K6: per-stream rate loop of the batched MP3 encoder for MI355X (gfx950).
// One workgroup of two wavefronts (a master that runs the encoder, a helper that takes channel 1's share
// and the traffic with HBM) owns one stream and walks its frames in order, because the
// allocator state (long-term MNR, per-band gain estimators, bit reservoir, scfsi memory)
// is carried frame to frame (reference bitallo3.cpp:484-3149, mp3enc.cpp:1492-1597,
// :2106-2333, l3pack.c:107-1187, bitalloc.cpp:470-811).
// This source is compiled five times: k_alloc and k_alloc_slim (MPEG-1; 256 registers and 38 KB of LDS per
// stream, four streams per CU - or 168 registers and 26 KB, six per CU: HX_SLIM, hx_alloc_slim.hip),
// k_alloc_lsf (MPEG-2 rates), k_alloc1 / k_alloc1_lsf (the reference's first-generation allocator).
//
// Inside a granule the work is spread over the lanes three ways:
//   line-parallel   quantise / x^(3/4) / M-S butterflies / noise terms / Huffman lengths
//                   (lane j handles lines j, j+64, ...);
//   band-parallel   every scalefactor band owns lane 32*ch + sfb for its integer control
//                   (noise targets, gain search state machine, scalefactors);
//   wave collective integer reductions and scans (bit counts, code offsets, maxima).
// Floating-point sums that the reference accumulates line by line are formed from
// line-parallel terms that the owning band lane then adds in the reference's order, so every
// decision (they are integer compares on millibel values) is bit-identical to the oracle.
// The gain search of all 42 bands advances in the same sweep.
#include "hx_dev.h"

#ifndef HX_LSF
#define HX_LSF 0            // 1 when compiled as hx_alloc_lsf.hip: MPEG-2 LSF streams, one granule per frame
#endif
#ifndef HX_A1
#define HX_A1 0             // 1 when compiled as hx_alloc1*.hip: streams of the first-generation allocator (intensity stereo, dual channel)
#endif
#define GMIN_OFFSET 70
#define PART23 4021
#define NB 22

// The scalar part of HxParams that the rate loop reads, staged in LDS: the class pointer comes
// from a load, so every p->field on the global struct would be a vector memory round trip.
// Field names match HxParams.
struct AllocPrm {
    unsigned char head[4];
    int remainder, divisor, main_framebytes, sf_bit_max, AveTargetBits;
    int ms_flag, hf_flag, vbr_flag;
    int ivbr_min, ivbr_max, vbr_main_framebytes[16], vbr_pool_target;
    int fbmin, fbmax;                   // main-data capacity of the lowest / highest VBR bitrate index
    int initialMNR, test1;
    int nsf[2], nsf2[2], nsf3[2], nbmax[2], nbmax2[2], nbmax3[2];
    float rnBand_l[22];
    int nsfs, nbmax_s;
    struct { int npart; } psyS;
    int nchan, side_bytes;              // 1 / 17 for a mono stream (mode 3), 2 / 32 otherwise
    int is_flag, dual, npart_l;         // first-generation allocator: intensity part present, dual channel; long psy partitions
    int oflags;                         // optional outputs of the call: 1 = packets, 2 = debug taps, 4 = per-frame counters, 8 = strict band sums everywhere
    int run_w;                          // lines per lane of the certified band sums (HxParams::run_w)
};

// What the master wave hands to global memory in the frame loop - the k_pack records of a granule's channels and,
// when the granule ends a frame, the frame's header, side information, record and slot - collects here and is
// stored by the helper wave a granule later.  A store of the master's own would stand in its in-order memory queue
// ahead of every later load (scratch reloads included), which then waits a memory round trip for the write's
// acknowledgement.  Two boxes: granule g fills box g & 1 while the helper empties the other.
static_assert(sizeof(HxGr) == 96, "the frame loop copies the four HxGr of a frame as 96 words");
struct alignas(16) Outbox {
    HxSegOut seg[2];
    HxGr gr[2][2];                      // the frame's side information as the allocator left it, its scfsi bits and
    int scfsi[2], mdb, part;            // main_data_begin (MPEG-2: which granule): the helper wave builds the bits
    unsigned sidew[10];                 // side information, MSB-first words (the helper's staging)
    unsigned char head[4];
    HxFrameOut frm;
    HxSlot slot;
    int opos, frm_index, slot_index;    // where they go: header offset in the stream's output, frame and slot number
    int ring_p0;                        // position of the frame's first slot in the ring of pending frames (r_off / r_mf)
    int has_frame;
};

// HX_SLIM = 1: the low-footprint layout of the stream walk's LDS (26.5 KB instead of 38.3 KB per stream: six workgroups per
// CU instead of four, for batches with more streams than the chip holds at once).  What differs from the layout it is
// derived from (all of it bit-identical in its results):
//  * the quantised lines are int16 and live in the first half of the noise-term buffer: a granule's terms (gain search,
//    big_lucky) are dead when its lines are quantised and the other way round; inverse_sf2 puts its squares into xr
//    (in place) and x34, both dead by then, instead of into the term buffer; the quantiser writes every line (zeros past
//    the coded range), since the buffer holds terms before it;
//  * the next granule's band start values land in the second half of the term buffer (the order to fetch them goes out
//    when the lines are final, and they are picked up before the next granule's first term is written);
//  * a short granule's signs are kept as bits while it is allocated and end up, in bitstream order, in the x34 buffer;
//  * gain tables, the x^(3/4) exponent table: 2^(k/4) and 2^(-3k/16) have period 4 / 16 in the mantissa, the tables are
//    ldexp of 4 / 16 constants (hx_host.cpp checks that identity on the host's tables before a batch may use this kernel);
//    the mB-log and log-subtract tables as 16-bit values;
//  * the short-block allocator's gain-step arrays and the band tables as int16, the scalefactor outputs as bytes, big_lucky's
//    work list in an array of its own.
#ifndef HX_SLIM
#define HX_SLIM 0
#endif
#if HX_SLIM && HX_A1
#error "the low-footprint layout is built for the second-generation allocator only"
#endif
#if HX_SLIM
typedef short ix_t;                     // a quantised line
typedef short sgs_t;                    // short-block gain steps / scalefactors per band (0 .. 127, flags 0 / -1)
typedef signed char sfo_t;              // a scalefactor as transmitted (0 .. 15; the first-generation allocator's -3 .. 999 never get here)
typedef short btab_t;                   // band tables: widths, first lines, log widths in mB, target tapers
#else
typedef int ix_t;
typedef int sgs_t;
typedef int sfo_t;
typedef int btab_t;
#endif

// The frame loop's carried scalars - bit reservoir, ring positions of the pending frames, layout cursor, running totals - and
// the budget of the frame being coded.  They live here and are read where a frame is budgeted and where it is placed: as
// register variables they occupied some sixty scalar registers over all of the allocator's code in between, which the compiler
// kept in lanes of three VGPRs - and, at the low-footprint build's 168 registers, spilled those to scratch (61 reload sites).
struct FrameState {
    int padcount; unsigned main_tot, main_sent, mf_tot;
    int main_bytes; unsigned side_p0, side_p1, tot_frames_out, tot_bytes_out; int ave_tot;
    int opos, slot_lo, slot_hi;
    int pad, byte_pool, byte_max, byte_min, bytesout, pk_first;     // the frame in flight
};

struct alignas(16) AllocLds {
    float xr[2][576];
    float x34[2][576];
    float term[2][576];
#if !HX_SLIM
    int ix[2][576];
    unsigned char signx[2][576];
    alignas(16) HxBandPrep band_next;               // the next granule's band start values, landed by LDS-DMA while this granule's are still in use
#else
    unsigned long long sgnbits[2][9];               // short blocks: sign of line t of channel c = bit t & 63 of word t >> 6
    ix_t ixsink;                                    // where the quantiser's store of a line it must leave alone goes
#endif
    unsigned char band_of_pair[288];                // sfb of lines 2 j, 2 j + 1 (bands start on even lines and have even widths)
    unsigned short lane_run[64];                    // the lanes' line runs for the certified band sums (HxParams::lane_run)
    unsigned char band_last[24];                    // last lane of each band's run of lanes
    // tables staged from global memory
    float look_ix43[256];
#if !HX_SLIM
    float look_gain[128], look_34igain[128];
    int mblog[256];
    float pow34_exp[256];
    int logsub[84];
#else
    float gain4[4], igain16[16];                    // look_gain[8 .. 11], look_34igain[8 .. 23]: the tables' mantissa periods
    unsigned short mblog[256];                      // the table without its constant -38227
    short logsub[84];
#endif
    float pow34_a[16], pow34_b[16], quant_off[32];
    unsigned char huff_len[1408];
    unsigned char sband_of_line[192];
    btab_t nBand_s[16], startBand_s[16], logcbw_s[16];
    unsigned char quada_len[16];
    btab_t nBand[NB], startBand[24], logcbw[NB], taper[NB];
    int NTadjust[2][NB];                // long-block gain estimator feedback (persists)
    union {
        struct {    // long blocks: per band working set, [channel][sfb]
            int snr[2][NB], Noise0[2][NB], Noise[2][NB], NT[2][NB];
            int gzero[2][NB], gmin[2][NB], gsf[2][NB], sf[2][NB], active[2][NB];
            int ixmax[2][NB], ix10xmax[2][NB], up[2][NB], lo[2][NB], geval[2][NB], maskmb[2][NB];
            float xsxx[2][NB], x34max[2][NB];
            float gig[2][NB];                   // 1 / gain^(3/4) of the band's quantiser step
            alignas(8) float2 gpair[2][NB];     // gain pair (1 / gain^(3/4), gain) of the band's current evaluation step: one read per line
            int lucky[6][2][13];                // big_lucky_noise: noise of candidate c of band (ch, sfb)
#if HX_SLIM
            unsigned short llist[6 * 26 + 4];   // big_lucky_noise: its work list
#endif
        };
        struct {    // first-generation allocator (hx_alloc1.inc): psy model output and noise / mask levels per band in dB
            int a_pad[17][2][NB];       // (the long-block arrays up to x34max stay in use)
            float a_sig[2][NB], a_smask[2][NB], a_mask[2][NB], a_noise[2][NB];
            int a_lastGsf[2][NB];
        };
        struct {    // short blocks: [channel][window][sfb]
            int s_snr[2][3][16], s_Noise0[2][3][16], s_Noise[2][3][16], s_NT[2][3][16];
            sgs_t s_gzero[2][3][16], s_gmin[2][3][16], s_gsf[2][3][16], s_sf[2][3][16], s_active[2][3][16];
            int s_ixmax[2][3][16], s_geval[3][16], s_tmpn[3][16], s_maskmb[2][3][16];
            float s_xsxx[2][3][16], s_x34max[2][3][16];
            int s_G[2][3], s_GG[2], s_subgain[2][3];
        };
    };
    // per channel
    int G[2], preemp[2], scale[2], huff_bits[2];
    int hs_table[2][4], hs_cbreg[2][3], hs_nbig[2], hs_nquads[2], hs_bits[2];
    // stream scalars (persist across frames)
    int MNR, PoolFraction, call_count;
    int a_calls, a_bitadjust[2];        // first-generation allocator's carried scalars
    float a_running, a_ave, a_alpha;
    int hf_quant, hf_quant_stereo[2], gsf_hf, gsf_hf_stereo[2];
    int sf_save[2][21];
    int scfsi[2];
    // call scalars
    int nchan, block_type, maxBits, maxTargetBits, minTargetBits, PoolBits, TargetBits, deltaMNR, activeBands;
    int tmp[8];
    int tmpn[2][NB];
    unsigned int sidew[10];             // bit staging of the side information
    HxGr gr[2][2];
    sfo_t sfout[2][2][NB];
    sfo_t sfs[2][3][12];                // short-block scalefactors of the current granule
    AllocPrm P;
    FrameState fs;
    int tabpk[32];                      // per Huffman table: code offset | row stride << 12 | linbits << 20
    unsigned long long candpk[19];      // candidate tables per class of a region's largest value
    unsigned short r_mf[32];            // pending frames: main-data bytes of the slot ...
    int r_off[32];                      // ... and offset of its header in the output buffer
#if !HX_SLIM
    float dump[64];                     // per-lane sink for predicated-off stores (keeps hot loops branch-free)
#endif
    Outbox ob[2];
    const double *pow43;                // HxGlobalTabs::pow43 (global memory)
    int *big_counter;                   // device counter of line passes that took the double table (tests)
    int nstrict;                        // certified band sums of this stream that fell back to the strict sum (added to big_counter[1] when the stream retires)
    int cur_s;                          // the stream this workgroup is walking (the helper wave reads it with a fetch order)
    alignas(16) int cmdw[4];            // work order for the helper wave (see HELPER_POST): command + three arguments, one 16-byte read
    alignas(4) unsigned char gflag[HX_SLIM ? 64 : 256];     // block type | stereo decision << 2 of the next granules (frame loop, hx_alloc3.inc)
#ifdef HX_PROFILE
    unsigned prof[64];                  // (the profile build holds three workgroups per CU instead of four: per-stream cycles are what it is for)
#endif
};

// Layout-dependent accessors (see HX_SLIM above)
#if HX_SLIM
#define IX(c) (reinterpret_cast<ix_t *>(&L.term[0][0]) + 576 * (c))
#define BAND_LANDING (reinterpret_cast<HxBandPrep *>(&L.term[1][0]))
#define LANE_SINK (reinterpret_cast<float *>(&L.lucky[0][0][0]) + LANE)       // big_lucky's results are written after its terms
#define MBLOG(x) hx_mblog16(L.mblog, (x))
// 2^((g - 8) / 4) and 2^(-3 (g - 8) / 16) as the tables hold them: float(2^(r / 4)) x 2^q is exact, and so is the period-16 form
// of the second (hx_host.cpp: slim_tables_ok)
__device__ __forceinline__ float lk_gain(const float *g4, int g) { const int k = g - 8; return ldexpf(g4[k & 3], k >> 2); }
__device__ __forceinline__ float lk_igain(const float *ig16, int g) { const int k = g - 8; return ldexpf(ig16[k & 15], -3 * (k >> 4)); }
#define LK_GAIN(g) lk_gain(L.gain4, (g))
#define LK_IGAIN(g) lk_igain(L.igain16, (g))
// the same in two halves: the table read (asked for early), the scaling (where the value is needed)
#define LK_GAIN_RAW(g) L.gain4[((g) - 8) & 3]
#define LK_GAIN_FIN(raw, g) ldexpf((raw), ((g) - 8) >> 2)
#define LK_IGAIN_RAW(g) L.igain16[((g) - 8) & 15]
#define LK_IGAIN_FIN(raw, g) ldexpf((raw), -3 * (((g) - 8) >> 4))
#else
#define IX(c) (&L.ix[(c)][0])
#define BAND_LANDING (&L.band_next)
#define LANE_SINK (&L.dump[LANE])
#define MBLOG(x) hx_mblog(L.mblog, (x))
#define LK_GAIN(g) L.look_gain[(g)]
#define LK_IGAIN(g) L.look_34igain[(g)]
#define LK_GAIN_RAW(g) L.look_gain[(g)]
#define LK_GAIN_FIN(raw, g) (raw)
#define LK_IGAIN_RAW(g) L.look_34igain[(g)]
#define LK_IGAIN_FIN(raw, g) (raw)
#endif
#define BAND_OF_LINE(j) L.band_of_pair[(j) >> 1]

#ifdef HX_PROFILE
// (only the master wave's time is booked: the helper wave runs some of the same functions)
#define PROF(id, stmt) do { SYNC(); long long t0_ = clock64(); stmt; SYNC(); if (threadIdx.x == 0 && (id) < 64) L.prof[(id) % 64] += (unsigned) (clock64() - t0_); } while (0)
#define PROF_T0() long long tp_ = clock64()
#define PROF_T1() tp_ = clock64()
#define PROF_CNT(id) do { if (threadIdx.x == 0 && (id) < 64) L.prof[(id) % 64] += 1; } while (0)
#define PROF_ACC(id) do { SYNC(); if (threadIdx.x == 0 && (id) < 64) L.prof[(id) % 64] += (unsigned) (clock64() - tp_); tp_ = clock64(); } while (0)
#else
#define PROF(id, stmt) do { stmt; } while (0)
#define PROF_T0() do { } while (0)
#define PROF_T1() do { } while (0)
#define PROF_CNT(id) do { } while (0)
#define PROF_ACC(id) do { } while (0)
#endif
// Rarely taken paths are kept out of line, away from the hot code: the kernel's instructions do not fit
// the instruction cache that the waves of a CU share (DESIGN.md, K6 in detail).
#define HX_COLD __attribute__((noinline, cold))
// The -HF helpers: out of line (most configurations never call them), but with -HF on they run in every granule: for speed.
#define HX_HFN __attribute__((noinline))
// The rate loop's correction paths (increase_bits, decrease_bits, limit_bits and the requantise-and-count they share) run in a
// minority of granules, but a launch ends with its slowest stream and that stream is one that lives in them: out of line like
// the cold functions (their code stays away from the common path's), but compiled for speed, not - as `cold` implies - for size.
// (Measured: config 2 +2.2 %, its worst-case signal set +2.4 %.  The 168-register build keeps them cold: compiled for speed they
// pull their callees in, the kernel grows by 6 KB and its scalar spills by 55: config 3 -2.2 %.)
#ifndef HX_RATE_SPEED
#define HX_RATE_SPEED (!HX_SLIM)
#endif
#if HX_RATE_SPEED
#define HX_RATE __attribute__((noinline))
#else
#define HX_RATE HX_COLD
#endif
// The lane number as a value the compiler cannot see through, taken once per function (HX_LANE_DECL at its top; LANE is that
// local).  As the pure threadIdx.x & 63 every predicate and LDS address derived from it - i < NB, LANE < 44, base + 4 * lane,
// dozens of them after inlining - is loop-invariant, gets hoisted out of the frame loop and stays live over all of it: some
// 130 scalar registers (lane predicates are register pairs) spilled into VGPR lanes, and at the 168-register build those VGPRs
// and 27 more into scratch, reloaded at 61 places of the frame loop.  Recomputed where a function starts, they cost two
// instructions there and die with the function.
__device__ __forceinline__ int hx_lane_opaque()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// (HX_OPAQUE_LANE: on for the 168-register build; the 256-register builds keep the plain value - there the hoisted values
// stay in registers, and recomputing them costs the stream's chain 2 %)
#ifndef HX_OPAQUE_LANE
#define HX_OPAQUE_LANE HX_SLIM
#endif
#if HX_OPAQUE_LANE
#define HX_LANE_DECL const int lane_ = hx_lane_opaque()
#else
#define HX_LANE_DECL const int lane_ = (int) threadIdx.x & 63
#endif
#define LANE lane_
#define WAVE ((int) threadIdx.x >> 6)
// The workgroup is a single wavefront, and a wave's LDS operations execute in issue order, so an
// LDS hand-over between lanes only needs the compiler to keep the accesses in program order.
// __syncthreads() would also drain every outstanding global load/store (s_waitcnt vmcnt(0)),
// which costs a memory round trip per call in the frame-level code.  SYNC_G() is the full
// barrier, used where lanes exchange data through global memory.
// (No s_waitcnt: the LDS unit takes a wave's DS instructions in issue order, so a read issued behind a write of the same
// wave sees it whichever lane wrote; waiting for the write's completion first only adds its latency - measured 1-2 % of
// the launch.  The compiler places the waits that register results need.)
#define SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#define SYNC_G() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// A stream's workgroup is two wavefronts.  Wave 0 (the master) runs the encoder; wave 1 (the helper)
// sleeps at the workgroup barrier until the master hands it one channel's share of a phase whose
// channels are independent: HELPER_POST publishes the order in LDS and releases the helper,
// HELPER_JOIN waits for it.  Everything else in this file is wave-local and never uses s_barrier.
// Workgroup barrier with the LDS hand-over around it: own LDS writes done before, and no LDS read
// of the other wave's data moved above it by the compiler (the s_barrier builtin alone does not
// order memory accesses).
#define WG_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
enum { HCMD_EXIT = 0, HCMD_COUNT_BITS, HCMD_QUANT, HCMD_ISF2, HCMD_LUCKY, HCMD_SEEK, HCMD_FETCH, HCMD_QUANT_COUNT, HCMD_SYNC /* nothing: the helper is through with what went before */ };
#define HELPER_POST(c_, a0_) do { if (LANE == 0) { L.cmdw[0] = (c_); L.cmdw[1] = (a0_); } \
        WG_BARRIER(); } while (0)
#define HELPER_POST2(c_, a0_, a1_) do { if (LANE == 0) { L.cmdw[0] = (c_); L.cmdw[1] = (a0_); L.cmdw[2] = (a1_); } \
        WG_BARRIER(); } while (0)
#define HELPER_JOIN() WG_BARRIER()

// 16 bytes per lane from global memory straight into LDS: lane l's bytes land at lds + 16 l (the LDS base is
// wave-uniform); completion is counted by vmcnt like any vector memory load
__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g, (__attribute__((address_space(3))) void *) lds, 16, 0, 0);
}

// x^(3/4): piecewise-linear mantissa fit x exponent table (reference pow34.c:132-154)
#if HX_SLIM
// (the exponent table 2^(3 (e - 127) / 4) as ldexp of the four mantissas 2^(r / 4); its ends are 0 and infinity)
__device__ __forceinline__ float pow34(const AllocLds &L, float x)
{
    const unsigned u = hx_f2bits(x);
    const float m = hx_bits2f((u & 0x7FFFFFu) | (127u << 23));
    const unsigned seg = (u >> 19) & 15;
    const int e = (int) ((u >> 23) & 255), k = 3 * (e - 127);
    float ex = ldexpf(L.gain4[k & 3], k >> 2);
    if (e == 0) ex = 0.0f;
    if (e == 255) ex = hx_bits2f(0x7F800000u);
    return (m * L.pow34_b[seg] + L.pow34_a[seg]) * ex;
}
#else
__device__ __forceinline__ float pow34(const AllocLds &L, float x) { return hx_pow34(L.pow34_a, L.pow34_b, L.pow34_exp, x); }
#endif

__device__ __forceinline__ int logsubber(const AllocLds &L, int n1, int n2)
{
    int k = (n1 - n2) >> 4;
    if (k > 83) k = 83;
    return n1 + L.logsub[k];
}

__device__ __forceinline__ int drop_guard(int noise0, int nt)
{
    int tsnr = noise0 - nt;
    if (tsnr < 300) { tsnr = 187 + ((3 * tsnr) >> 3) - tsnr; nt -= tsnr; }
    return nt;
}

// scalefactor limits by [scalefac_scale][preflag] (reference bitallo3.cpp:87-162)
__device__ __forceinline__ int pretab(int i)
{
    return (i < 11 || i > 20) ? 0 : (i < 15 ? 1 : (i < 17 ? 2 : (i < 20 ? 3 : 2)));
}
__device__ __forceinline__ int sf_limit_hi(int scale, int pre, int i)
{
    int base = (i < 11) ? (scale ? 62 : 31) : (scale ? 30 : 15);
    if (pre) base += (scale ? 4 : 2) * pretab(i);
    return base;
}
__device__ __forceinline__ int sf_limit_lo(int scale, int pre, int i) { return pre ? (scale ? 4 : 2) * pretab(i) : 0; }
__device__ __forceinline__ int sf_upper(int scale, int pre, int i)
{
    int base = (i < 11) ? (scale ? 60 : 30) : (scale ? 28 : 14);
    if (pre) base += (scale ? 4 : 2) * pretab(i);
    return base;
}

// ---------------------------------------------------------------------------------------
// Noise sweep: every band with geval >= 0 gets its quantisation noise (mB) measured at gain
// step geval (reference l3math.c:512-541).  Result left in L.tmp-free per-band array `out`.
// one line of a noise measurement: (x - gain * ix^(4/3))^2 for ix = round(igain * x34 - 0.0946)
// (reference l3math.c:521-535).  *fast is false when ix falls outside the 256-entry table.
__device__ __forceinline__ float noise_term(const AllocLds &L, float igain, float gain, float x34, float x, bool *fast)
{
    float tmp = (igain * x34 + (0.0f - 0.0946f));
    const int qx = (int) (tmp + copysignf(0.5f, tmp));
    *fast = (qx >= 0 && qx < 256);
    const float xhat = gain * L.look_ix43[*fast ? qx : 0];
    tmp = x - xhat;
    return tmp * tmp;
}
// The reference's value beyond the float table: gain * pow(ix, 4/3) in double.  ix^(4/3) comes from the double table
// in global memory (host pow(), as the reference calls it); pow() itself only past the table's 16384 entries.
__device__ __noinline__ double pow43_beyond(int qx) { return pow((double) qx, (4.0 / 3.0)); }
__device__ __forceinline__ float noise_xhat_big(const AllocLds &L, int qx, double pw_tab, float gain)
{
    double pw = pw_tab;
    if (qx >= HX_POW43_N) pw = pow43_beyond(qx);
    return (float) (gain * pw);
}

// Table-only variant for the hot loops: the index is clamped into the table, and the caller has
// established per band (from the band's largest x^(3/4), the index is monotone in it) that no
// line needs the pow() path - or repairs the band's lines afterwards.
// (The reference rounds half away from zero, tmp + copysign(0.5, tmp).  Here tmp >= -0.0946 for every line that is read
// afterwards - igain > 0, x34 >= 0 - and on [-0.0946, 0) both tmp - 0.5 and tmp + 0.5 truncate to 0: one instruction less.)
__device__ __forceinline__ float noise_term_fast(const AllocLds &L, float igain, float gain, float x34, float x)
{
    float tmp = (igain * x34 + (0.0f - 0.0946f));
    const unsigned qx = (unsigned) (int) (tmp + 0.5f);
    const float xhat = gain * L.look_ix43[min(qx, 255u)];
    tmp = x - xhat;
    return tmp * tmp;
}
__device__ __forceinline__ bool noise_band_needs_pow(float igain, float x34max)
{
    const float tmp = (igain * x34max + (0.0f - 0.0946f));
    return (int) (tmp + copysignf(0.5f, tmp)) >= 256;
}

// Line operands of the gain search, held in registers for a whole seek_actual call: lane l owns a run of up to run_w
// consecutive lines of one band (x and x^(3/4); hx_dev.h, "certified band sums"); lines past the run's end are zeros,
// whose noise term is +0.  (one channel per wave: the master keeps channel 0, the helper wave channel 1)
#define RUNW_MAX 10
struct SweepRegs { float x34[RUNW_MAX], xr[RUNW_MAX]; };
struct LaneRun { int start, cnt, d, band; };    // this lane's run: first line, lines, lanes back to the band's first lane, the band

__device__ __forceinline__ LaneRun lane_run(const AllocLds &L)
{
    HX_LANE_DECL;
    const unsigned r = L.lane_run[LANE];
    LaneRun q;
    q.start = (int) (r & 511u) << 1; q.cnt = (int) ((r >> 9) & 7u) << 1; q.d = (int) (r >> 12);
    q.band = L.band_of_pair[r & 511u];
    return q;
}

__device__ __forceinline__ void sweep_load(const AllocLds &L, SweepRegs &R, const LaneRun &q, int W, int ch)
{
#pragma unroll
    for (int k = 0; k < RUNW_MAX; k += 2) {
        float2 a = make_float2(0.0f, 0.0f), b = make_float2(0.0f, 0.0f);
        if (k < W) {
            // (a run never ends inside a pair; a read past the run stays inside the LDS block and is dropped)
            a = *reinterpret_cast<const float2 *>(&L.x34[ch][q.start + k]);
            b = *reinterpret_cast<const float2 *>(&L.xr[ch][q.start + k]);
            if (k >= q.cnt) { a = make_float2(0.0f, 0.0f); b = a; }
        }
        R.x34[k] = a.x; R.x34[k + 1] = a.y; R.xr[k] = b.x; R.xr[k + 1] = b.y;
    }
}

// The lane's share of a band's noise: terms of its run for the band's published gain pair, added up (tree of pairs).
__device__ __forceinline__ float sweep_run(const AllocLds &L, const SweepRegs &R, const LaneRun &q, int W, int ch)
{
    const float2 gp = L.gpair[ch][q.band];
    float t[RUNW_MAX];
#pragma unroll
    for (int k = 0; k < RUNW_MAX; k += 2) {
        t[k] = t[k + 1] = 0.0f;
        if (k < W) {
            t[k] = noise_term_fast(L, gp.x, gp.y, R.x34[k], R.xr[k]);
            t[k + 1] = noise_term_fast(L, gp.x, gp.y, R.x34[k + 1], R.xr[k + 1]);
        }
    }
    return (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) + (t[8] + t[9]);
}

// The same from LDS, with the terms also stored in line order (L.term) for a lane that has to add a band strictly, and
// terms beyond the 256-entry table taken from the double table (loud near-mono material at low gain steps; decided per
// sweep from the band maxima).  Out of line: the rare cases cost the common path neither registers nor code.
// (Bands that are not evaluated quantise to <= 0; count: book the pass in the double-table counter.)
__device__ __noinline__ float sweep_run_stored(AllocLds &L, int ch, int W, int count)
{
    HX_LANE_DECL;
    if (count && LANE == 0) atomicAdd(L.big_counter, 1);
    const LaneRun q = lane_run(L);
    const float2 gp = L.gpair[ch][q.band];
    float acc = 0.0f;
#pragma unroll 1
    for (int k = 0; k < W; k += 2) {
        float t[2], xr[2];
        int qx[2];
        const float2 a = *reinterpret_cast<const float2 *>(&L.x34[ch][q.start + k]);
        const float2 b = *reinterpret_cast<const float2 *>(&L.xr[ch][q.start + k]);
        const bool in = k < q.cnt;
        const float x34[2] = {in ? a.x : 0.0f, in ? a.y : 0.0f};
        xr[0] = in ? b.x : 0.0f; xr[1] = in ? b.y : 0.0f;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float tmp = (gp.x * x34[e] + (0.0f - 0.0946f));
            qx[e] = (int) (tmp + copysignf(0.5f, tmp));
            t[e] = noise_term_fast(L, gp.x, gp.y, x34[e], xr[e]);
        }
        // (a pass none of whose lines quantises beyond the float table - the loud lines sit in a few low bands - does not
        // go to the double table at all: its memory round trip is most of this function)
        if (__any(qx[0] >= 256 || qx[1] >= 256)) {
            double pw[2];
#pragma unroll
            for (int e = 0; e < 2; e++) pw[e] = L.pow43[min(max(qx[e], 0), HX_POW43_N - 1)];
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (qx[e] >= 256) { const float d = xr[e] - noise_xhat_big(L, qx[e], pw[e], gp.y); t[e] = d * d; }
        }
        if (in) *reinterpret_cast<float2 *>(&L.term[ch][q.start + k]) = make_float2(t[0], t[1]);
        acc += (t[0] + t[1]);
    }
    return acc;
}

// A band whose certified interval straddles a bucket boundary of mbLogC (a few per cent of the sweeps have one): its lane
// adds the band's terms in line order.  have_terms: the terms are in L.term already (the double-table pass stores them).
__device__ __noinline__ float sweep_sum_strict(AllocLds &L, int ch, int W, int have_terms, bool need, int sbeg, int n, float fast)
{
    HX_LANE_DECL;
    if (!have_terms) (void) sweep_run_stored(L, ch, W, 0);
    SYNC();
    if (LANE == 0) atomicAdd(&L.nstrict, 1);
    float sxx = fast;
    if (need) sxx = band_sum(&L.term[ch][sbeg], n, 0.0f);
    return sxx;
}

// One sweep of the gain search for channel ch, run by one wave on its own (the master wave does
// channel 0 while the helper wave does channel 1: gain pairs, terms and sums of the two channels live
// in separate LDS arrays, so the waves never wait for each other inside a search).  Band lane i < 32
// passes the gain step g it wants measured (-1: none) and gets the band's noise back in a register;
// sbeg / send are the lane's band limits, kept by the caller.
// (ig, gn = the step's gain pair 1 / gain^(3/4), gain: the caller reads the tables a sweep ahead, see seek_actual_ch;
// x34max, logcbw = the band lane's constants, kept by the caller; q, W: the lane's run; du, last4: the band lane's interval
// half-width and 4 x the last lane of its band's run of lanes)
__device__ __forceinline__ int noise_sweep(AllocLds &L, const SweepRegs &R, const LaneRun &q, int W, int ch, int g, float ig_g, float gn_g, float x34max, int logcbw, int sbeg, int send, float du, int last4)
{
    HX_LANE_DECL;
    // band lanes publish the gain pair of their evaluation step (igain < 0: band not evaluated)
    bool bslow = false;
    PROF_T0();
    if (LANE < NB) {
        const float ig = (g >= 0) ? ig_g : -1.0f;
        L.gpair[ch][LANE] = make_float2(ig, (g >= 0) ? gn_g : 0.0f);
        if (g >= 0) bslow = noise_band_needs_pow(ig, x34max);
    }
    SYNC();
    PROF_ACC(27);
    const int anyslow = __any(bslow) ? 1 : 0;
    float part;
    if (__builtin_expect(anyslow, 0)) part = sweep_run_stored(L, ch, W, 1);
    else part = sweep_run(L, R, q, W, ch);
    PROF_ACC(28);
    // the band's total arrives in its last lane; the band lane fetches it and certifies the bucket
    float sxx = hx_lane_read(last4, hx_seg_scan(part, q.d, LANE));
    bool strict = false;
    if (g >= 0) strict = du < 0.0f || !hx_cert_mblog(sxx, du);
    if (__builtin_expect(__any(strict), 0)) sxx = sweep_sum_strict(L, ch, W, anyslow, strict, sbeg, send - sbeg, sxx);
    int noise = 0;
    if (g >= 0) noise = MBLOG(1.0e-12f + sxx) - logcbw;
    SYNC();
    PROF_ACC(29);
    return noise;
}

// ---------------------------------------------------------------------------------------
// reference bitallo3.cpp:1069-1126
__device__ void adjust_nt(AllocLds &L, const AllocPrm *p)
{
    HX_LANE_DECL;
    const int f = p->test1;
    if (f == 0) return;
    const int ch = LANE >> 5, i = LANE & 31;
    const int nsf = p->nsf[ch];
    const int sth = (i < 14) ? 0 : (i < 17 ? 100 : (i == 17 ? 200 : 300));
    const bool sel = (i < nsf) && (L.snr[ch][i] > sth);
    // integer sums within each half-wave (channel)
    int na = sel ? 1 : 0, ab = sel ? L.nBand[i] * L.NT[ch][i] : 0, nab = sel ? L.nBand[i] : 0;
    na = hx_half_sum(na); ab = hx_half_sum(ab); nab = hx_half_sum(nab);
    na += 1; nab += 1;
    ab = ab / nab;
    if (na >= 5 && sel) {
        int dmax = max(L.snr[ch][i] - 400, 0);
        int d = (f * (ab - L.NT[ch][i])) >> 4;
        d = min(d, dmax);
        L.NT[ch][i] = L.NT[ch][i] + d;
    }
    SYNC();
}

// Start of a long-block granule (reference bitallo3.cpp:816-898 L/R, :902-1066 M/S).  The parts that do not
// depend on the stream's carried state - magnitudes, signs, band energies, x^(3/4), band maxima, zero-gain
// steps, masking thresholds - were computed by k_prep (hx_front.hip) and arrive in `in`; what is left here are
// the noise targets, which follow the long-term MNR: band-parallel integer work.
struct BandIn { float xsxx, x34max; int n0, n0ms, gzero, maskmb; };     // band lane (ch, sfb)'s share of HxBandPrep

__device__ __forceinline__ BandIn band_fetch(const HxBandPrep *bp)
{
    HX_LANE_DECL;
    const int ch = LANE >> 5, i = min(LANE & 31, NB - 1);
    BandIn b;
    b.xsxx = bp->xsxx[ch][i]; b.x34max = bp->x34max[ch][i]; b.n0 = bp->n0[ch][i]; b.n0ms = bp->n0ms[ch][i];
    b.gzero = bp->gzero[ch][i]; b.maskmb = bp->maskmb[ch][i];
    return b;
}

__device__ void startup_prepped(AllocLds &L, const AllocPrm *p, int ms, const BandIn &in)
{
    HX_LANE_DECL;
    if (ms) {
        if (LANE == 0 && p->vbr_flag == 0 && L.call_count > 10 && (L.TargetBits - L.minTargetBits) < 100)
            L.MNR = min(L.MNR + 50, 2050);
        SYNC();
    }
    const int mnr = ms ? L.MNR : L.MNR + 100;
    const int ch = LANE >> 5, i = LANE & 31;
    const bool eband = i < (ms ? p->nsf[0] : p->nsf3[ch]);          // bands whose energy the reference forms
    const bool band = i < (ms ? p->nsf[0] : p->nsf[ch]);
    const int nbz = ms ? p->nsf2[ch] : p->nsf3[ch];                 // bands that get a zero-gain step
    int act = 0;
    if (i < NB) L.x34max[ch][i] = in.x34max;
    if (eband) L.xsxx[ch][i] = in.xsxx;
    if (band) {
        const int cbw = L.logcbw[i], n0 = in.n0;
        int nt;
        if (n0 < -2000) nt = ms ? 10000 : n0 + 1000;
        else { act = L.nBand[i]; nt = drop_guard(n0, (in.maskmb - cbw) - mnr + L.taper[i]); }
        L.NT[ch][i] = nt;
        L.snr[ch][i] = n0 - nt;
        L.Noise0[ch][i] = ms ? in.n0ms : n0;
    }
    if (i < nbz) { L.gzero[ch][i] = in.gzero; L.gmin[ch][i] = max(0, in.gzero - GMIN_OFFSET); }
    act = hx_wave_sum(act);
    if (LANE == 0) L.activeBands = act;
    SYNC();
    adjust_nt(L, p);
    if (ms && LANE < p->nsf[0]) {       // targets of M and S from those of L and R (bitallo3.cpp:1014-1062)
        int b = LANE;
        int NTL = L.NT[0][b], NTR = L.NT[1][b], Nsum = L.Noise0[0][b], Ndiff = L.Noise0[1][b];
        int xNT = min(NTL, NTR) + 300, nt0, nt1;
        nt0 = nt1 = xNT;
        if (Ndiff < xNT) { nt0 = logsubber(L, xNT, Ndiff); if (b < 16) nt0 -= 200; }
        if (Nsum < xNT) nt1 = logsubber(L, xNT, Nsum);
        L.NT[0][b] = nt0; L.NT[1][b] = nt1;
        L.snr[0][b] = Nsum - nt0;
        L.snr[1][b] = Ndiff - nt1;
    }
    SYNC();
}

// reference bitallo3.cpp:1130-1160
__device__ void seek_initial(AllocLds &L, const AllocPrm *p)
{
    HX_LANE_DECL;
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < p->nsf[ch]) {
        int na = L.NTadjust[ch][i];
        na = max(na, -400);
        na = min(na, 400);
        L.NTadjust[ch][i] = na;
        float g4 = 0.017716950f * MBLOG(L.x34max[ch][i]) + (88.411238f - 100.0f + 8.0f);
        float d = (1.00f / 110.5f) * (1800 - 8 * i - (L.Noise0[ch][i] - L.NT[ch][i] + na));
        float g = g4 + d;
        int gs = hx_round(g);
        gs = min(gs, L.gzero[ch][i]);
        gs = max(gs, L.gmin[ch][i]);
        L.gsf[ch][i] = gs;
    }
    SYNC();
}

// reference bitallo3.cpp:1164-1296: all bands of channel ch walk their gain step concurrently (lane = sfb)
#ifndef HX_SEEK_FORCEINLINE
#define HX_SEEK_FORCEINLINE 0
#endif
#if HX_SEEK_FORCEINLINE
#define HX_SEEK_INLINE __device__ __forceinline__
#else
#define HX_SEEK_INLINE __device__
#endif
HX_SEEK_INLINE void seek_actual_ch(AllocLds &L, const AllocPrm *p, int ch)
{
    HX_LANE_DECL;
    const int i = LANE;
    const bool band = i < p->nsf[ch];
    // per-lane state machine: mode 0 = idle/done, 1 = first measurement, 2 = walking down, 3 = walking up.
    // The lane's band data (target, limits, estimator feedback, result) stays in registers over the
    // sweeps and is written back once.
    int mode = 0, s = 0, t = 0, NTarget = 0, absmin = 0, tnmin = 0, smin = 0, iter = 0, niter = 0;
    int ntadj = 0, sbeg = 0, send = 0;
    if (band) {
        NTarget = L.NT[ch][i];
        s = L.gsf[ch][i];
        ntadj = L.NTadjust[ch][i];
        sbeg = L.startBand[i];
        send = sbeg + L.nBand[i];
        if (L.Noise0[ch][i] > NTarget) mode = 1;
        else { smin = L.gzero[ch][i] + 5; tnmin = L.Noise0[ch][i]; }
    }
    SweepRegs R;
    const int W = p->run_w;
    const LaneRun q = lane_run(L);
    sweep_load(L, R, q, W, ch);
    // The gain pair of a step comes from two tables in LDS.  A walking band's next step is known before the current
    // one is measured (one down or one up), so its pair is read a sweep ahead and the sweep starts without that
    // round trip; the first measurement reads both neighbours.  (Indices are clamped for the reads only: a step
    // that would leave the table is never evaluated - niter / the gain limits stop the walk first.)
    const int ib = min(i, NB - 1);
    const float x34max = L.x34max[ch][ib];
    const int logcbw = L.logcbw[ib];
    // the band lane's interval half-width (strict sums everywhere: no interval passes) and where its band's total arrives
    const float du = (p->oflags & 8) ? -1.0f : hx_cert_delta(send - sbeg, W);
    const int last4 = 4 * (int) L.band_last[ib];
    float ig_c = LK_IGAIN(s & 127), gn_c = LK_GAIN(s & 127);      // pair of the step to measure now
    float ig_dn = 0.0f, gn_dn = 0.0f, ig_up = 0.0f, gn_up = 0.0f;          // pairs of the steps below / above it
    SYNC();
    while (__any(mode != 0)) {
        PROF_CNT(20);
#ifdef HX_PROFILE
        if (threadIdx.x == 64) L.prof[45] += 1;     // the helper wave's sweeps (channel 1)
#endif
        const int gcur = (mode == 0) ? -1 : (mode == 1 ? s : t);
        // requested now, used after this sweep (low-footprint layout: the tables' mantissas now, their scaling then)
        const int gdn = max(max(gcur, 0) - 1, 0), gup = min(max(gcur, 0) + 1, 127);
        if (mode != 3) { ig_dn = LK_IGAIN_RAW(gdn); gn_dn = LK_GAIN_RAW(gdn); }
        if (mode != 2) { ig_up = LK_IGAIN_RAW(gup); gn_up = LK_GAIN_RAW(gup); }
        const int noise = noise_sweep(L, R, q, W, ch, gcur, ig_c, gn_c, x34max, logcbw, sbeg, send, du, last4);
        if (mode == 1) {
            const int dn = noise - NTarget;
            ntadj += (dn >> 3);
            absmin = abs(dn); tnmin = noise; smin = s; iter = 0;
            if (dn > 100) { t = s - 1; niter = min(t, 20); mode = (niter > 0) ? 2 : 0; ig_c = LK_IGAIN_FIN(ig_dn, gdn); gn_c = LK_GAIN_FIN(gn_dn, gdn); }
            else if (dn < -100) { t = s + 1; niter = 20; mode = 3; ig_c = LK_IGAIN_FIN(ig_up, gup); gn_c = LK_GAIN_FIN(gn_up, gup); }
            else mode = 0;
        } else if (mode == 2 || mode == 3) {
            const int ad = abs(noise - NTarget);
            if (ad < absmin) { absmin = ad; tnmin = noise; smin = t; }
            iter++;
            const bool stop = (mode == 2) ? (noise <= NTarget) : (noise >= NTarget);
            if (stop || iter >= niter) mode = 0;
            else if (mode == 2) { t -= 1; ig_c = LK_IGAIN_FIN(ig_dn, gdn); gn_c = LK_GAIN_FIN(gn_dn, gdn); }
            else { t += 1; ig_c = LK_IGAIN_FIN(ig_up, gup); gn_c = LK_GAIN_FIN(gn_up, gup); }
        }
    }
    if (band) { L.gsf[ch][i] = smin; L.Noise[ch][i] = tnmin; L.NTadjust[ch][i] = ntadj; }
    if (i < NB) L.geval[ch][i] = -1;
    SYNC();
}

// both channels: channel 1's whole search on the helper wave
__device__ void seek_actual(AllocLds &L, const AllocPrm *p)
{
    HX_LANE_DECL;
    const bool two = p->nbmax[1] > 0;
    if (two) HELPER_POST(HCMD_SEEK, 0);
    seek_actual_ch(L, p, 0);
    PROF_T0();
    if (two) HELPER_JOIN();
    else if (LANE < NB) L.geval[1][LANE] = -1;
    SYNC();
    PROF_ACC(30);
}

// ---------------------------------------------------------------------------------------
// Global gain / scalefactors (reference bitallo3.cpp:1793-1862, 1892-2019 L/R, 2022-2170 M/S).
// Band-parallel: channel ch lives in lanes 32*ch .. 32*ch+21; maxima / ORs are half-wave
// reductions.  The M/S variant carries the running maximum from a silent channel 0 into
// channel 1 (the reference only resets it on the non-silent path).
__device__ __forceinline__ int half_max(int v) { return hx_half_max(v); }
__device__ __forceinline__ int half_or(int v) { return hx_half_or(v); }

__device__ int scale_factors(AllocLds &L, const AllocPrm *p, int ms)
{
    HX_LANE_DECL;
    const int ch = LANE >> 5, i = LANE & 31;
    const bool band = i < p->nsf[ch];
    int gsf = 0, gz = 0, act = 0;
    if (band) {
        gsf = max(L.gsf[ch][i], L.gmin[ch][i]);
        gz = L.gzero[ch][i];
        act = (gsf < gz) ? -1 : 0;
    }
    int gact = half_max((band && act) ? gsf : -1);          // max over active bands, -1 if none
    int gzmax = half_max(band ? gz : -1);
    int g0init, g1init;
    if (ms) {
        g0init = L.hf_quant ? L.gsf_hf : -1;
        int G0 = max(g0init, __builtin_amdgcn_readlane(gact, 0));
        // channel 0 silent -> its Gtmp (max of gzero, and of the initial value) leaks into channel 1
        int leak = max(G0, __builtin_amdgcn_readlane(gzmax, 0));
        g1init = (G0 < 0) ? leak : -1;
    } else {
        g0init = L.gsf_hf_stereo[0];
        g1init = L.gsf_hf_stereo[1];
    }
    int Gtmp = max(ch ? g1init : g0init, gact);
    const bool silent = Gtmp < 0;
    int sf = 0, pre = 0, scale = 0, dsf = 2;
    if (silent) {
        Gtmp = max(Gtmp, gzmax);
        if (band) { sf = 0; gsf = gz; }
    } else {
        if (band) sf = (Gtmp - gsf) & act;
        int sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0;
        if (band && act) {
            sp0 = (sf_limit_hi(0, 0, i) - sf);
            sp1 = (sf_limit_hi(0, 1, i) - sf) | (sf - sf_limit_lo(0, 1, i));
            sp2 = (sf_limit_hi(1, 0, i) - sf);
            sp3 = (sf_limit_hi(1, 1, i) - sf) | (sf - sf_limit_lo(1, 1, i));
        }
        sp0 = half_or(sp0); sp1 = half_or(sp1); sp2 = half_or(sp2); sp3 = half_or(sp3);
        if (sp0 >= 0) { scale = 0; pre = 0; }
        else if (sp1 >= 0) { scale = 0; pre = 1; }
        else if (sp2 >= 0) { scale = 1; pre = 0; }
        else if (sp3 >= 0) { scale = 1; pre = 1; }
        else { scale = 1; pre = 0; }
        if (HX_LSF && sp0 < 0) { scale = 1; pre = 0; }     // fnc_sf_final_MPEG2 (bitallo3.cpp:1860-1888): no pre-emphasis
        if (band) {
            int noise = L.Noise[ch][i], nt = L.NT[ch][i];
            if (scale == 0) {
                dsf = 2;
                if (ms) {
                    if (act) {
                        if ((gz - gsf) < 5) sf++;
                        else if ((i < 11) && (noise > nt)) sf++;
                        sf &= (~1);
                    }
                } else {
                    if ((i < 11) && (noise > nt)) sf++;
                    sf &= (~1);
                }
            } else {
                dsf = 4;
                if (!ms || act) {
                    int s = sf & (~3), d = sf - s;
                    int dN = noise - nt + 150 * d;
                    int thr = (i < 15) ? 250 : (i == 15 ? 300 : (i < 18 ? 400 : (i < 20 ? 500 : 600)));
                    if (dN > thr) s = s + 4;
                    else if (ms && (gz - gsf - d) < 5) s = s + 4;
                    sf = ms ? s : (s & act);
                }
            }
        }
    }
    if (i < NB) {
        int up = sf_upper(silent ? 0 : scale, silent ? 0 : pre, i), lo = sf_limit_lo(silent ? 0 : scale, silent ? 0 : pre, i);
        L.up[ch][i] = up;
        L.lo[ch][i] = lo;
        if (band && !silent) {
            if (sf > up) sf = up; else if (sf < lo) sf = lo;
            if (act) {
                gsf = Gtmp - sf;
                if (gsf < 0) { gsf += dsf; sf -= dsf; }
                if (gsf >= gz) { gsf = gz + 5; sf = lo; }
            }
        }
        if (band) { L.gsf[ch][i] = gsf; L.sf[ch][i] = sf; L.active[ch][i] = silent ? 0 : act; }
    }
    if (i == 0) { L.G[ch] = Gtmp; L.preemp[ch] = pre; L.scale[ch] = scale; }
    SYNC();
    return 0;
}

// Noise terms of up to K candidates for the lines of sfb 0..12 of both channels, flattened to
// t = ch * nl + line (sfb 0..12 end at line 88 / 90 / 102 at 48 / 44.1 / 32 kHz, so NQ = 3 or 4
// slots of 64 lanes).  A lane keeps the operands of its slots in registers over the candidates;
// three candidates x NQ slots are independent LDS chains per block, all loads ahead of the
// stores.  Terms of (candidate, band) pairs that are not being measured are computed and stored
// too - nobody reads them, and it keeps the loop free of predicates.
// A wave takes every other slot: w = 0 the even ones (master), w = 1 the odd ones (helper wave).
template <int NQ>
__device__ __forceinline__ void lucky_terms(AllocLds &L, int nl, int ncmax, float *tf, int w)
{
    HX_LANE_DECL;
    float sx34[NQ], sxr[NQ];
    int sg[NQ], ssd[NQ], stride[NQ];
    float *base[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int t = LANE + 64 * (2 * q + w);
        const bool ok = t < 2 * nl;
        const int cc = (ok && t >= nl) ? 1 : 0, j = ok ? t - (cc ? nl : 0) : 0;
        sx34[q] = L.x34[cc][j];
        sxr[q] = L.xr[cc][j];
        sg[q] = max(L.geval[cc][BAND_OF_LINE(j)], 0);
        ssd[q] = 2 * (1 + L.scale[cc]);
        base[q] = ok ? &tf[t] : LANE_SINK;
        stride[q] = ok ? 2 * nl : 0;
    }
    for (int c0 = 0; c0 < ncmax; c0 += 3) {
        float v[3][NQ];
#pragma unroll
        for (int cu = 0; cu < 3; cu++)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int g = min(sg[q] + (c0 + cu) * ssd[q], 127);
                v[cu][q] = noise_term_fast(L, LK_IGAIN(g), LK_GAIN(g), sx34[q], sxr[q]);
            }
#pragma unroll
        for (int cu = 0; cu < 3; cu++)
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (c0 + cu < ncmax) base[q][(c0 + cu) * stride[q]] = v[cu][q];
    }
}

// The same with terms beyond the 256-entry table taken from the double table (see sweep_lines_big): out of line, one
// candidate at a time.
template <int NQ>
__device__ __noinline__ void lucky_terms_big(AllocLds &L, int nl, int ncmax, float *tf, int w)
{
    HX_LANE_DECL;
    float sx34[NQ], sxr[NQ];
    int sg[NQ], ssd[NQ], stride[NQ];
    float *base[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int t = LANE + 64 * (2 * q + w);
        const bool ok = t < 2 * nl;
        const int cc = (ok && t >= nl) ? 1 : 0, j = ok ? t - (cc ? nl : 0) : 0;
        sx34[q] = L.x34[cc][j];
        sxr[q] = L.xr[cc][j];
        sg[q] = max(L.geval[cc][BAND_OF_LINE(j)], 0);
        ssd[q] = 2 * (1 + L.scale[cc]);
        base[q] = ok ? &tf[t] : LANE_SINK;
        stride[q] = ok ? 2 * nl : 0;
    }
#pragma unroll 1
    for (int c = 0; c < ncmax; c++) {
        float v[NQ], gn[NQ];
        int qx[NQ];
        double pw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int g = min(sg[q] + c * ssd[q], 127);
            const float ig = LK_IGAIN(g);
            gn[q] = LK_GAIN(g);
            const float tmp = (ig * sx34[q] + (0.0f - 0.0946f));
            qx[q] = (int) (tmp + copysignf(0.5f, tmp));
            v[q] = noise_term_fast(L, ig, gn[q], sx34[q], sxr[q]);
        }
        bool anybig = false;
#pragma unroll
        for (int q = 0; q < NQ; q++) anybig = anybig || qx[q] >= 256;
        if (__any(anybig)) {        // (a candidate none of whose lines passes the float table skips the double table's round trip)
#pragma unroll
            for (int q = 0; q < NQ; q++) pw[q] = L.pow43[min(max(qx[q], 0), HX_POW43_N - 1)];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (qx[q] >= 256) { const float d = sxr[q] - noise_xhat_big(L, qx[q], pw[q], gn[q]); v[q] = d * d; }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) base[q][c * stride[q]] = v[q];
    }
}

__device__ __forceinline__ void lucky_dispatch(AllocLds &L, int nl, int ncmax, float *tf, int w, int big)
{
    if (__builtin_expect(big, 0)) {
        if (!HX_LSF || 2 * nl <= 256) lucky_terms_big<2>(L, nl, ncmax, tf, w);
        else lucky_terms_big<3>(L, nl, ncmax, tf, w);
    } else {
        if (!HX_LSF || 2 * nl <= 256) lucky_terms<2>(L, nl, ncmax, tf, w);
        else lucky_terms<3>(L, nl, ncmax, tf, w);
    }
}

// reference bitallo3.cpp:1348-1396
__device__ void big_lucky_noise(AllocLds &L, const AllocPrm *p)
{
    HX_LANE_DECL;
    // The candidates of a band (scalefactor s, s - sdelta, ... while G - s stays below gzero - 4)
    // do not depend on each other's result, so up to K of them are measured per pass for all
    // bands at once; the band lane then replays the reference's scan over the results in order.
    const int ch = LANE >> 5, i = LANE & 31;
    const int m = min(13, p->nsf[ch]);
    const int nl = L.startBand[13];                         // lines of sfb 0..12
    const int K = min(6, 1152 / (2 * nl));
    float *tf = &L.term[0][0];                               // [K][2][nl]
    int mode = 0, s = 0, s0 = 0, g0 = 0, GG = 0, sdelta = 2, smin = 0, nt = 0;
    if (i < m && L.active[ch][i] && (L.gsf[ch][i] < (L.gzero[ch][i] - 5))) {
        sdelta = 2 * (1 + L.scale[ch]);
        GG = L.G[ch];
        smin = L.sf[ch][i];
        g0 = L.gzero[ch][i] - 4;
        s = min(L.sf[ch][i] - sdelta, L.up[ch][i]);
        s0 = L.lo[ch][i];
        nt = L.NT[ch][i];
        mode = 1;
        if (!(s >= s0) || (GG - s) >= g0) mode = 2;        // loop body never runs
    }
    while (__any(mode == 1)) {
        PROF_CNT(21);
        PROF_T0();
        // valid candidates of this pass: the longest prefix c = 0.. with s - c*sdelta >= s0 and
        // G - s + c*sdelta < g0 (sdelta is 2 or 4)
        int nc = 0;
        if (mode == 1) {
            const int sh = 1 + L.scale[ch];
            nc = min(K, min(((s - s0) >> sh) + 1, (g0 - GG + s + sdelta - 1) >> sh));
        }
        if (i < NB) { L.geval[ch][i] = (mode == 1) ? GG - s : -1; L.tmpn[ch][i] = nc; }
        // work list of the sums: one entry (c, ch, sfb) per candidate, compacted over the band lanes
#if HX_SLIM
        unsigned short *list = L.llist;
#else
        int *list = IX(0);                            // ix is not live before do_quant
#endif
        const int incl = hx_wave_scan(nc), total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
        for (int c = 0; c < 6; c++) if (c < nc) list[incl - nc + c] = (c << 8) | (ch << 7) | i;
        SYNC();
        const int ncmax = hx_wave_max(nc);
        const bool bslow = mode == 1 && noise_band_needs_pow(LK_IGAIN(GG - s), L.x34max[ch][i]);
        PROF_ACC(23);
        // slots of 64 flattened lines: 3 or 4 (MPEG-2 band tables: 5), shared between the two waves
        const bool two = p->nchan == 2;
        const int big = __any(bslow) ? 1 : 0;       // a band reaches beyond the 256-entry table
        if (two) { if (LANE == 0) L.cmdw[3] = big; HELPER_POST2(HCMD_LUCKY, nl, ncmax); }
        lucky_dispatch(L, nl, ncmax, tf, 0, big);
        if (two) HELPER_JOIN();
        else lucky_dispatch(L, nl, ncmax, tf, 1, big);
        SYNC();
        PROF_ACC(24);
        for (int u = LANE; u < total; u += 64) {
            const int e = list[u], c = e >> 8, cc = (e >> 7) & 1, b = e & 31;
            float sxx = band_sum(tf + c * 2 * nl + cc * nl + L.startBand[b], L.nBand[b], 0.0f);
            L.lucky[c][cc][b] = MBLOG(1.0e-12f + sxx) - L.logcbw[b];
        }
        SYNC();
        PROF_ACC(25);
        {   // replay the reference's scan: the last candidate that meets the target wins
            int nz[6];
#pragma unroll
            for (int c = 0; c < 6; c++) nz[c] = L.lucky[c][ch][min(i, 12)];
            int best = 0;
            bool hit = false;
#pragma unroll
            for (int c = 0; c < 6; c++)
                if (c < nc && nz[c] <= nt) { best = nz[c]; smin = s - c * sdelta; hit = true; }
            if (hit) L.Noise[ch][i] = best;
            if (mode == 1) {
                s -= nc * sdelta;
                if (!(s >= s0) || (GG - s) >= g0) mode = 2;
            }
        }
        SYNC();
        PROF_ACC(26);
    }
    if (mode == 2) {
        L.sf[ch][i] = smin;
        L.gsf[ch][i] = max(GG - smin, 0);
    }
    // The work list lived in channel 0's line buffer (at most 6 x 26 entries).  The quantiser rewrites the coded lines
    // only: with a very low subband limit (E_CONTROL.nsb_limit = 4 at 48 kHz: 72 lines) the list's tail would stay behind as lines
    // (found by the round-3 sweep with nsb_limit in the draw).  Lines past the coded range are zero by contract.
    // (the low-footprint layout keeps the list elsewhere, and its quantiser writes every line)
    if (!HX_SLIM) { for (int j = p->nbmax[0] + LANE; j < 6 * 26; j += 64) IX(0)[j] = 0; }
    SYNC();
}

// reference bitallo3.cpp:1540-1585 with l3math.c:656-694: quantise every coded band
// quantise channel c's lines with the band gains published in L.gig, track the band maxima
__device__ __forceinline__ void quant_lines(AllocLds &L, const AllocPrm *p, int opt, int c)
{
    HX_LANE_DECL;
    // all nine lines of a lane in one basic block: band -> igain -> rounding offset are dependent LDS reads, the nine
    // chains overlap; stores (and the band maxima) come after all loads
    const int nl = p->nbmax[c];
#if HX_SLIM
    const int nfill = (opt & 2) ? nl : 576;
#endif
    int q[9], b[9];
#pragma unroll
    for (int k = 0; k < 9; k++) b[k] = BAND_OF_LINE(LANE + 64 * k);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int j = LANE + 64 * k;
        const float igain = L.gig[c][b[k]];
        if (opt & 1) {
            float t = igain * L.x34[c][j] + (0.5f - 0.4375f);
            int iq = (int) t;
            if (iq > 31) iq = 31;
            q[k] = (int) (t - L.quant_off[iq < 0 ? 0 : iq]);
        } else {
            q[k] = (int) (igain * L.x34[c][j] + (0.5f - 0.0946f));
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int j = LANE + 64 * k;
#if HX_SLIM
        // the line buffer held noise terms before: every line is written, zeros past the coded range - unless the buffer
        // holds this granule's lines already (opt & 2: the L/R rate loop's step towards more bits): there the reference
        // leaves what an earlier -HF pass put into band 21, and its count of the last quadruples can reach into it
        // (one value, one unconditional store: written as two conditions the compiler made two exec-masked paths per line)
        const int qq = (j < nl) ? q[k] : 0;
        *((j < nfill) ? &IX(c)[j] : &L.ixsink) = (ix_t) qq;
        if (qq > 0) atomicMax(&L.ixmax[c][b[k]], qq);
#else
        if (j < nl) {
            IX(c)[j] = q[k];
            if (q[k] > 0) atomicMax(&L.ixmax[c][b[k]], q[k]);
        }
#endif
    }
}

__device__ void do_quant(AllocLds &L, const AllocPrm *p, int opt)
{
    HX_LANE_DECL;
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < NB) {
        L.ixmax[ch][i] = (i < p->nsf[ch]) ? 0 : L.ixmax[ch][i];
        L.gig[ch][i] = LK_IGAIN(L.gsf[ch][i] & 127);     // the band's 1/gain^(3/4), read per line below
    }
    SYNC();
    // three lines per lane and chunk: band -> igain -> rounding offset are dependent LDS reads,
    // the three chains overlap; stores (and the band maximum) come after all loads of the chunk
    const bool two = p->nbmax[1] > 0;
    if (two) HELPER_POST(HCMD_QUANT, opt);          // channel 1 on the helper wave
    quant_lines(L, p, opt, 0);
    if (two) HELPER_JOIN();
    SYNC();
}

// ---------------------------------------------------------------------------------------
// Huffman region split, table choice and bit count for one channel
// (reference bitalloc.cpp:310-420, 470-756; cnt.c:96-325).
// Candidate Huffman tables of a region by its largest value (reference cnttab.h:38-62,
// bitalloc.cpp:310-420).  Scalars only: a runtime-indexed member array would live in scratch.
struct Cand { int n, t0, t1, t2, t3, tmax; };

__device__ __forceinline__ Cand mk_cand(int n, int t0, int t1, int t2, int t3, int tmax)
{
    Cand c; c.n = n; c.t0 = t0; c.t1 = t1; c.t2 = t2; c.t3 = t3; c.tmax = tmax;
    return c;
}

__device__ __forceinline__ Cand candidates_chain(int rmax)
{
    if (rmax <= 0) return mk_cand(0, 0, 0, 0, 0, 0);
    if (rmax == 1) return mk_cand(2, 1, 3, 0, 0, 1);
    if (rmax == 2) return mk_cand(2, 2, 3, 0, 0, 2);
    if (rmax == 3) return mk_cand(2, 5, 6, 0, 0, 3);
    if (rmax <= 5) return mk_cand(4, 7, 8, 9, 12, 5);
    if (rmax <= 7) return mk_cand(4, 10, 11, 12, 15, 7);
    if (rmax <= 15) return mk_cand(2, 13, 15, 0, 0, 15);
    if (rmax == 16) return mk_cand(2, 16, 24, 0, 0, 16);
    if (rmax <= 18) return mk_cand(2, 17, 24, 0, 0, 18);
    if (rmax <= 22) return mk_cand(2, 18, 24, 0, 0, 22);
    if (rmax <= 30) return mk_cand(2, 19, 24, 0, 0, 30);
    if (rmax <= 46) return mk_cand(2, 25, 20, 0, 0, 46);
    if (rmax <= 78) return mk_cand(2, 20, 26, 0, 0, 78);
    if (rmax <= 142) return mk_cand(2, 27, 21, 0, 0, 142);
    if (rmax <= 270) return mk_cand(2, 21, 28, 0, 0, 270);
    if (rmax <= 526) return mk_cand(2, 29, 22, 0, 0, 526);
    if (rmax <= 1038) return mk_cand(2, 22, 30, 0, 0, 1038);
    if (rmax <= 2062) return mk_cand(2, 30, 23, 0, 0, 2062);
    return mk_cand(2, 31, 23, 0, 0, 8206);
}

// The same through the per-class table staged in LDS (class = position of rmax among the
// thresholds 0,1,2,3,5,7,15 and 15 + 2^k): a handful of scalar ops and one read instead of the
// compare chain on every call.
__device__ __forceinline__ int cand_class(int rmax)
{
    if (rmax <= 3) return max(rmax, 0);
    if (rmax <= 15) return rmax <= 5 ? 4 : (rmax <= 7 ? 5 : 6);
    return min(7 + (31 - __clz(rmax - 15)), 18);
}
__device__ __forceinline__ int cand_class_rep(int c)     // smallest rmax of class c
{
    return c <= 3 ? c : (c == 4 ? 4 : (c == 5 ? 6 : (c == 6 ? 8 : 15 + (1 << (c - 7)))));
}
__device__ __forceinline__ unsigned long long cand_pack(const Cand &c)
{
    return (unsigned long long) (c.n | (c.t0 << 3) | (c.t1 << 8) | (c.t2 << 13) | (c.t3 << 18)) | ((unsigned long long) c.tmax << 23);
}
__device__ __forceinline__ Cand candidates(const AllocLds &L, int rmax)
{
    const unsigned long long pk = L.candpk[cand_class(rmax)];
    const unsigned lo = (unsigned) pk;
    return mk_cand(lo & 7, (lo >> 3) & 31, (lo >> 8) & 31, (lo >> 13) & 31, (lo >> 18) & 31, (int) (pk >> 23));
}

// table parameters packed into one word: code-table offset | row stride << 12 | linbits << 20
// (tables >= 16 are 16 x 16 with escapes above 14; the smaller ones have no escapes)
__device__ __forceinline__ int tab_pack(const AllocLds &L, int t) { return L.tabpk[t]; }
__device__ __forceinline__ int pair_len_p(const AllocLds &L, int pk, int x, int y)
{
    const int off = pk & 0xFFF, dim = (pk >> 12) & 0xFF, lin = pk >> 20;
    const int cx = min(x, 15), cy = min(y, 15);
    int n = L.huff_len[off + cx * dim + cy];
    n += (x >= 15 ? lin : 0) + (y >= 15 ? lin : 0);
    return n + (x != 0) + (y != 0);
}

// coded length of one pair in table t: Huffman length + sign bits + linbits
__device__ __forceinline__ int pair_len(const AllocLds &L, int t, int x, int y) { return pair_len_p(L, tab_pack(L, t), x, y); }

// packed 16-bit length sums of one pair for the candidates of a region
__device__ __forceinline__ void acc_pair(const AllocLds &L, const Cand &c, int x, int y, int &p01, int &p23)
{
    if (c.n == 0) return;
    p01 += pair_len(L, c.t0, x, y) | (pair_len(L, c.t1, x, y) << 16);
    if (c.n == 4) p23 += pair_len(L, c.t2, x, y) | (pair_len(L, c.t3, x, y) << 16);
}

// pick the shortest candidate; ties go to the higher index (reference cnt.c:45,111-120)
__device__ __forceinline__ int pick_table(const Cand &c, int p01, int p23, int live, int *bits)
{
    const int b0 = p01 & 0xFFFF, b1 = (p01 >> 16) & 0xFFFF, b2 = p23 & 0xFFFF, b3 = (p23 >> 16) & 0xFFFF;
    int best = 0, t = c.t0;
    if (c.n != 0 && live) {
        if (b0 < b1) { best = b0; t = c.t0; } else { best = b1; t = c.t1; }
        if (c.n == 4) {
            if (b2 <= best) { best = b2; t = c.t2; }
            if (b3 <= best) { best = b3; t = c.t3; }
        }
    }
    *bits += best;
    return t;
}

__device__ __forceinline__ int region_max(const int *ixmax, int a, int b)
{
    int m = 0;
    for (int i = a; i < b; i++) if (m < ixmax[i]) m = ixmax[i];
    return m;
}

__device__ int count_bits_ch(AllocLds &L, const AllocPrm *p, int ch, int ncb)
{
    HX_LANE_DECL;
    const int *ixmax = L.ixmax[ch];
    const ix_t *ix = IX(ch);
    const int bt = L.block_type;
    PROF_T0();
    int cb0, cb1, cb2, cb3;
    // lane i holds ixmax[i]; the band scans of the reference become ballots
    const int my = (LANE < ncb) ? ixmax[LANE] : 0;
    const unsigned long long mk0 = __ballot(my > 0), mk1 = __ballot(my > 1);
    cb3 = mk0 ? 64 - __clzll((long long) mk0) : 0;      // one past the last band with a non-zero line
    cb2 = mk1 ? 64 - __clzll((long long) mk1) : 0;      // one past the last band with a value > 1
    cb0 = cb1 = 0;
#define RMAX(a_, b_) hx_wave_max((LANE >= (a_) && LANE < (b_)) ? my : 0)
    if (bt == 0) { if (cb2 < 2) { cb2 = 2; if (cb3 < cb2) cb3 = cb2; } }
    else { cb0 = 8; cb2 = max(cb2, 8); cb3 = max(cb3, cb2); cb1 = cb0; }
    // topmost line > 1 in the last "big" band, topmost line > 0 in the last count1 band
    PROF_ACC(36);
    int lo2 = L.startBand[cb2 - 1], hi2 = L.startBand[cb2], lo3 = L.startBand[cb3 - 1], hi3 = L.startBand[cb3];
    int j2 = lo2, j3 = lo3;
    for (int j = lo2 + LANE; j < hi2; j += 64) if (ix[j] > 1) j2 = j;
    for (int j = lo3 + LANE; j < hi3; j += 64) if (ix[j] > 0) j3 = j;
    j2 = hx_wave_max(j2);
    j3 = hx_wave_max(j3);
    PROF_ACC(37);
    int nbig = (j2 + 2) & (~1);
    if (bt == 0) { if (nbig < L.startBand[2]) nbig = L.startBand[2]; }
    else { if (nbig < L.startBand[8]) nbig = L.startBand[8]; }
    int nquads = (j3 + 4 - nbig) >> 2;
    if (bt != 0) nquads = max(nquads, 0);
    Cand c0, c1, c2;
    if (bt == 0) {
        const int c = cb2;      // region_table (reference bitalloc.cpp:124-197)
        int r0 = (c < 6) ? 1 : (c < 9 ? 2 : (c < 12 ? 3 : (c < 15 ? 4 : (c < 18 ? 5 : (c < 21 ? 6 : 7)))));
        int r1 = (c < 5) ? 1 : (c < 8 ? 2 : (c < 10 ? 3 : (c == 10 ? 4 : (c < 15 ? 5 : (c < 17 ? 6 : (c < 20 ? 7 : 8))))));
        cb0 = r0;
        cb1 = r0 + r1;
        if (cb0 < 1) cb0 = 1;
        if (cb1 <= cb0) cb1 = cb0 + 1;
        if (cb1 > cb0 + 8) cb1 = cb0 + 8;
        c0 = candidates(L, RMAX(0, cb0));
        c1 = candidates(L, RMAX(cb0, cb1));
        c2 = candidates(L, RMAX(cb1, cb2));
        if (c2.tmax < c1.tmax) {        // shrink region 1: last band in (cb0, cb1) above region 2's table range
            const unsigned long long m = __ballot(my > c2.tmax && LANE > cb0 && LANE <= cb1 - 1);
            const int j = m ? 63 - __clzll((long long) m) : cb0;
            cb1 = j + 1;
        }
        if (c1.tmax < c0.tmax) {        // shrink region 0 (region 1 stays <= 8 bands)
            int n = cb1 - 8;
            if (n < 1) n = 1;
            if (cb0 - 1 > n) {
                const unsigned long long m = __ballot(my > c1.tmax && LANE > n && LANE <= cb0 - 1);
                const int j = m ? 63 - __clzll((long long) m) : n;
                cb0 = j + 1;
            }
        }
    } else {
        c0 = candidates(L, RMAX(0, cb0));
        c1 = candidates(L, 0);
        c2 = candidates(L, RMAX(cb0, cb2));
    }
#undef RMAX
    PROF_ACC(38);
    const int n0 = L.startBand[cb0], n1 = L.startBand[cb1];
    // pair lengths: region 0 = [0,n0), region 1 = [n0,n1), region 2 = [n1,nbig).
    // The table parameters of the three regions are wave-uniform and read once; a lane picks its
    // region's parameters with selects, so a pair costs one table read per candidate and the
    // five pairs of a lane are independent chains in one basic block.
    int r0a = 0, r0b = 0, r1a = 0, r1b = 0, r2a = 0, r2b = 0;
    {
        const int pa0 = tab_pack(L, c0.t0), pb0 = tab_pack(L, c0.t1);
        const int pa1 = tab_pack(L, c1.t0), pb1 = tab_pack(L, c1.t1);
        const int pa2 = tab_pack(L, c2.t0), pb2 = tab_pack(L, c2.t1);
        const bool any4 = c0.n == 4 || c1.n == 4 || c2.n == 4;
        int2 xy[5];
#pragma unroll
#if HX_SLIM
        for (int k = 0; k < 5; k++) { const unsigned w = reinterpret_cast<const unsigned *>(ix)[min(LANE + 64 * k, 287)]; xy[k] = make_int2((int) (w & 0xFFFFu), (int) (w >> 16)); }
#else
        for (int k = 0; k < 5; k++) xy[k] = reinterpret_cast<const int2 *>(ix)[min(LANE + 64 * k, 287)];
#endif
        int acc[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int j = 2 * (LANE + 64 * k);
            const int pa = (j < n0) ? pa0 : (j < n1 ? pa1 : pa2), pb = (j < n0) ? pb0 : (j < n1 ? pb1 : pb2);
            acc[k] = pair_len_p(L, pa, xy[k].x, xy[k].y) | (pair_len_p(L, pb, xy[k].x, xy[k].y) << 16);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int j = 2 * (LANE + 64 * k);
            const int rn = (j < n0) ? c0.n : (j < n1 ? c1.n : c2.n);
            const int v = (j < nbig && rn != 0) ? acc[k] : 0;
            r0a += (j < n0) ? v : 0;
            r1a += (j >= n0 && j < n1) ? v : 0;
            r2a += (j >= n1) ? v : 0;
        }
        if (any4) {     // regions whose largest value is 4..7 have four candidate tables
            const int qa0 = tab_pack(L, c0.t2), qb0 = tab_pack(L, c0.t3);
            const int qa1 = tab_pack(L, c1.t2), qb1 = tab_pack(L, c1.t3);
            const int qa2 = tab_pack(L, c2.t2), qb2 = tab_pack(L, c2.t3);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j = 2 * (LANE + 64 * k);
                const int pa = (j < n0) ? qa0 : (j < n1 ? qa1 : qa2), pb = (j < n0) ? qb0 : (j < n1 ? qb1 : qb2);
                acc[k] = pair_len_p(L, pa, xy[k].x, xy[k].y) | (pair_len_p(L, pb, xy[k].x, xy[k].y) << 16);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j = 2 * (LANE + 64 * k);
                const int rn = (j < n0) ? c0.n : (j < n1 ? c1.n : c2.n);
                const int v = (j < nbig && rn == 4) ? acc[k] : 0;
                r0b += (j < n0) ? v : 0;
                r1b += (j >= n0 && j < n1) ? v : 0;
                r2b += (j >= n1) ? v : 0;
            }
        }
    }
    PROF_ACC(39);
    int qa = 0, qb = 0;
    {
        int2 qv[3][2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int q = min(LANE + 64 * k, max(nquads - 1, 0));
#if HX_SLIM
            const uint2 w2 = *reinterpret_cast<const uint2 *>(ix + nbig + 4 * q);      // (nbig is even: 4-byte aligned)
            qv[k][0] = make_int2((int) (w2.x & 0xFFFFu), (int) (w2.x >> 16)); qv[k][1] = make_int2((int) (w2.y & 0xFFFFu), (int) (w2.y >> 16));
#else
            const int2 *v2 = reinterpret_cast<const int2 *>(ix + nbig + 4 * q);
            qv[k][0] = v2[0]; qv[k][1] = v2[1];
#endif
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int v0 = qv[k][0].x, v1 = qv[k][0].y, v2 = qv[k][1].x, v3 = qv[k][1].y;
            const int pop = v0 + v1 + v2 + v3;
            const int la = L.quada_len[((v0 << 3) + (v1 << 2) + (v2 << 1) + v3) & 15] + pop;
            const bool ok = LANE + 64 * k < nquads;
            qa += ok ? la : 0;
            qb += ok ? 4 + pop : 0;
        }
    }
    PROF_ACC(40);
    r0a = hx_wave_sum(r0a); r2a = hx_wave_sum(r2a);
    r0b = hx_wave_sum(r0b); r2b = hx_wave_sum(r2b);     // unconditional: the DPP chains interleave
    r1a = hx_wave_sum(r1a); r1b = hx_wave_sum(r1b);
    int bits = 0;
    const int tab0 = pick_table(c0, r0a, r0b, n0 > 0, &bits);
    int tab1 = pick_table(c1, r1a, r1b, (n1 - n0 > 0) && bt == 0, &bits);
    const int tab2 = pick_table(c2, r2a, r2b, nbig - n1 > 0, &bits);
    if (bt != 0) tab1 = tab2;
    {
        const int q = hx_wave_sum(qa | (qb << 16));
        qa = q & 0xFFFF;
        qb = (q >> 16) & 0xFFFF;
    }
    int qidx = 0;
    if (nquads > 0) { if (qa < qb) { bits += qa; qidx = 0; } else { bits += qb; qidx = 1; } }
    if (LANE == 0) {
        L.hs_table[ch][0] = tab0; L.hs_table[ch][1] = tab1; L.hs_table[ch][2] = tab2; L.hs_table[ch][3] = qidx;
        L.hs_cbreg[ch][0] = cb0; L.hs_cbreg[ch][1] = cb1; L.hs_cbreg[ch][2] = cb2;
        L.hs_nbig[ch] = nbig; L.hs_nquads[ch] = nquads; L.hs_bits[ch] = bits;
        L.huff_bits[ch] = bits;
    }
    PROF_ACC(41);
    return bits;
}

__device__ int count_bits(AllocLds &L, const AllocPrm *p, const int *ncb)
{
    HX_LANE_DECL;
    PROF_CNT(22);
    // the channels are counted at the same time: channel 1 by the helper wave
    const bool two = p->nchan == 2;
    if (two) HELPER_POST(HCMD_COUNT_BITS, ncb[1]);
    int bits = count_bits_ch(L, p, 0, ncb[0]);
    if (two) { HELPER_JOIN(); bits += L.hs_bits[1]; }
    SYNC();
    return bits;
}

// Quantise and count in one work order (do_quant followed by count_bits when no HF lines are quantised in between):
// each wave quantises its channel's lines and goes straight on to count them, one hand-over to the helper instead of two.
// zero21: clear the mid channel's entry of band 21 before counting (M/S granules, reference bitallo3.cpp:640-645).
__device__ int quant_count_bits(AllocLds &L, const AllocPrm *p, int opt, int zero21, const int *ncb)
{
    HX_LANE_DECL;
    const int ch = LANE >> 5, i = LANE & 31;
    if (i < NB) {
        L.ixmax[ch][i] = (i < p->nsf[ch]) ? 0 : L.ixmax[ch][i];
        L.gig[ch][i] = LK_IGAIN(L.gsf[ch][i] & 127);
    }
    SYNC();
    PROF_CNT(22);
    const bool two = p->nchan == 2;
    PROF_T0();
    if (two) HELPER_POST2(HCMD_QUANT_COUNT, opt, ncb[1]);
    PROF_ACC(42);
    quant_lines(L, p, opt, 0);
    SYNC();
    PROF_ACC(43);
    if (zero21) { if (LANE == 0) L.ixmax[0][21] = 0; SYNC(); }
    int bits = count_bits_ch(L, p, 0, ncb[0]);
    PROF_T1();
    if (two) { HELPER_JOIN(); bits += L.hs_bits[1]; }
    SYNC();
    PROF_ACC(44);
    return bits;
}

#include "hx_alloc2.inc"
#include "hx_alloc_short.inc"
#if HX_A1
#include "hx_alloc1.inc"
#endif
#include "hx_alloc3.inc"
