// hx_pack.hip - K7: bitstream packing of the batched MP3 encoder for MI355X (gfx950), one workgroup per frame.
//
// The allocator kernel (hx_alloc*.hip) walks a stream's frames in order because its state is carried from frame
// to frame; writing the bits is not part of that chain.  It leaves, per (granule, channel), the quantised
// lines, the scalefactor fields and the Huffman regions, with the bit position where the segment starts in its
// frame's main data (known from the bit counts), and per frame where the main data goes in the output.  Here
// every frame of the batch is packed at the same time: wave w of a frame's workgroup writes segment w
// (granule w / 2, channel w % 2) into an LDS bit buffer - code words built five pairs per lane, placed with a
// wave prefix sum of their lengths, OR-ed in with LDS atomics - and the workgroup then moves the bytes to their
// place in the pending frames' slots (reference l3pack.c:157-558 scalefactors, :946-1119 Huffman codes,
// mp3enc.cpp:2258-2325 frame assembly).
#include "hx_dev.h"

struct alignas(16) PackLds {
    unsigned bitw[640];                 // the frame's main data, MSB-first 32-bit words
    unsigned short ix[4][576];          // per segment: quantised magnitudes ...
    unsigned sg[4][HX_SGN_WORDS];       // ... and signs, one bit per line
    unsigned short huff_code[1408];
    unsigned char huff_len[1408];
    int tabpk[32];                      // per Huffman table: code offset | row stride << 12 | linbits << 20
    unsigned char quada_code[16], quada_len[16];
    unsigned short sf[4][40];           // the frame's scalefactor fields (kept for unmasked_writer_fixup)
    int seghdr[4][8];                   // ... and its segment records
    int negflag;                        // some scalefactor of the frame is negative
};

// OR an n-bit field (n <= 32) at absolute bit position pos
__device__ __forceinline__ void put_bits(PackLds &L, int pos, unsigned val, int n)
{
    if (n <= 0) return;
    const int w = pos >> 5, o = pos & 31;
    const unsigned long long v = ((unsigned long long) val) << (64 - n - o);    // field left-aligned in 64 bits at offset o
    const unsigned hi = (unsigned) (v >> 32), lo = (unsigned) v;
    if (hi) atomicOr(&L.bitw[w], hi);
    if (lo) atomicOr(&L.bitw[w + 1], lo);
}
__device__ __forceinline__ void put_bits64(PackLds &L, int pos, unsigned long long val, int n)
{
    if (n > 32) { put_bits(L, pos, (unsigned) (val >> 32), n - 32); put_bits(L, pos + n - 32, (unsigned) val, 32); }
    else put_bits(L, pos, (unsigned) val, n);
}

// Huffman-code one segment; returns the new bit position.  Phase A builds the code word of every pair (five per
// lane, table parameters of the three regions read once and picked with selects: independent chains, no
// branches); phase B places them with a wave prefix sum of the lengths per 64 pairs.  Then the count1 quads.
__device__ __forceinline__ int pack_huff(PackLds &L, int pos, unsigned r01, unsigned r2q, unsigned tabs, const unsigned short *ix, const unsigned *sgn, int lane)
{
    const int n0 = r01 & 0xFFFF, n1 = r01 >> 16, n2 = r2q & 0xFFFF;
    const int npairs = n0 + n1 + n2;
    const int pk0 = L.tabpk[tabs & 31], pk1 = L.tabpk[(tabs >> 8) & 31], pk2 = L.tabpk[(tabs >> 16) & 31];
    unsigned long long val[5];
    int len[5];
    {
        unsigned xy[5], sn[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int pc = min(lane + 64 * k, 287);
            xy[k] = reinterpret_cast<const unsigned *>(ix)[pc];
            sn[k] = (sgn[pc >> 4] >> (2 * (pc & 15))) & 3u;         // the pair's two sign bits
        }
        int o[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int pi = lane + 64 * k;
            const int pk = (pi < n0) ? pk0 : (pi < n0 + n1 ? pk1 : pk2);
            o[k] = (pk & 0xFFF) + min((int) (xy[k] & 0xFFFF), 15) * ((pk >> 12) & 0xFF) + min((int) (xy[k] >> 16), 15);
        }
        unsigned code[5], hl[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { code[k] = L.huff_code[o[k]]; hl[k] = L.huff_len[o[k]]; }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int pi = lane + 64 * k;
            const int pk = (pi < n0) ? pk0 : (pi < n0 + n1 ? pk1 : pk2);
            const int dim = (pk >> 12) & 0xFF, lin = pk >> 20, x = (int) (xy[k] & 0xFFFF), y = (int) (xy[k] >> 16);
            unsigned long long v = code[k];
            int l = (int) hl[k];
            const int lx = (x >= 15) ? lin : 0, ly = (y >= 15) ? lin : 0;       // escapes (lin = 0 below table 16)
            v = (v << lx) | (unsigned) ((x >= 15) ? x - 15 : 0);
            l += lx;
            if (x) { v = (v << 1) | (sn[k] & 1u); l++; }
            v = (v << ly) | (unsigned) ((y >= 15) ? y - 15 : 0);
            l += ly;
            if (y) { v = (v << 1) | ((sn[k] >> 1) & 1u); l++; }
            val[k] = v;
            len[k] = (pi < npairs && dim != 0) ? l : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (64 * k < npairs) {
            const int incl = hx_wave_scan(len[k]);
            if (len[k]) put_bits64(L, pos + incl - len[k], val[k], len[k]);
            pos += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    const int nq = (int) (r2q >> 16), qb = 2 * npairs;
    const int c1sel = (int) (tabs >> 24);
    unsigned qval[3];
    int qlen[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int q = lane + 64 * k, qc = min(q, max(nq - 1, 0));
        const unsigned *v2 = reinterpret_cast<const unsigned *>(ix + qb + 4 * qc);
        const unsigned a = v2[0], b = v2[1];
        const int sp = qb + 4 * qc;                 // (even: the quad's four bits lie in one word or straddle two)
        const unsigned s4 = (unsigned) ((((unsigned long long) sgn[min((sp >> 5) + 1, HX_SGN_WORDS - 1)] << 32) | sgn[sp >> 5]) >> (sp & 31)) & 15u;
        const int code = (int) (((a & 0xFFFF) << 3) + ((a >> 16) << 2) + ((b & 0xFFFF) << 1) + (b >> 16));
        unsigned v;
        int l;
        if (c1sel == 1) { v = code ^ 15; l = 4; }
        else { v = L.quada_code[code & 15]; l = L.quada_len[code & 15]; }
        if (code & 8) { v = (v << 1) | (s4 & 1u); l++; }
        if (code & 4) { v = (v << 1) | ((s4 >> 1) & 1u); l++; }
        if (code & 2) { v = (v << 1) | ((s4 >> 2) & 1u); l++; }
        if (code & 1) { v = (v << 1) | ((s4 >> 3) & 1u); l++; }
        qval[k] = v;
        qlen[k] = (q < nq) ? l : 0;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (64 * k < nq) {
            const int incl = hx_wave_scan(qlen[k]);
            if (qlen[k]) put_bits(L, pos + incl - qlen[k], qval[k], qlen[k]);
            pos += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    return pos;
}

// The reference's bit writer does `bitbuf = (bitbuf << n) | x` without masking x (l3pack.c bitput), and the
// first-generation allocator can hand it a negative scalefactor: its upper bits then land on the bits written before
// it that have not left the 32-bit buffer yet.  Which those are depends on the writer's flush state, i.e. on the size of
// every single put since the frame's main data began (a put that does not fit flushes whole bytes until at least 24
// bits are free).  Rare (a few frames in a thousand dual-channel / intensity streams), so one lane replays the frame's
// sequence of put sizes - scalefactor fields, then per pair code / linbits / sign / linbits / sign, then the quads -
// and ORs the stray bits in where a negative field turns up.  The fields' own low bits are already in place.
__device__ __noinline__ void unmasked_writer_fixup(PackLds &L, int nseg)
{
    int P = 0;          // bits in the writer's buffer
    auto put = [&](int n) { if (32 - P < n) P = ((P - 1) & 7) + 1; P += n; };
    for (int w = 0; w < nseg; w++) {
        const int *h = L.seghdr[w];
        if (!h[7]) continue;                    // not this stream's segment (mono) 
        int pos = h[0];
        for (int j = 0; j < 40; j++) {
            const int fld = L.sf[w][j];
            if (!fld) continue;
            const int n = (fld >> 8) & 15;
            if (32 - P < n) P = ((P - 1) & 7) + 1;
            if (fld & 0x8000) {
                const int x = (int) (signed char) (fld & 255);
                const unsigned stray = (unsigned) (x >> n) & (P >= 32 ? 0xFFFFFFFFu : ((1u << P) - 1u));
                if (P > 0) put_bits(L, pos - P, stray, P);
            }
            P += n; pos += n;
        }
        if (!h[5]) continue;                    // no Huffman data
        const unsigned r01 = (unsigned) h[2], r2q = (unsigned) h[3], tabs = (unsigned) h[4];
        const int n0 = r01 & 0xFFFF, n1 = r01 >> 16, n2 = r2q & 0xFFFF, npairs = n0 + n1 + n2;
        const unsigned short *ix = L.ix[w];
        for (int pi = 0; pi < npairs; pi++) {
            const int pk = L.tabpk[(pi < n0) ? (tabs & 31) : (pi < n0 + n1 ? ((tabs >> 8) & 31) : ((tabs >> 16) & 31))];
            const int dim = (pk >> 12) & 0xFF, lin = pk >> 20;
            if (dim == 0) continue;
            const int x = ix[2 * pi], y = ix[2 * pi + 1];
            put(L.huff_len[(pk & 0xFFF) + min(x, 15) * dim + min(y, 15)]);
            if (x >= 15 && lin) put(lin);
            if (x) put(1);
            if (y >= 15 && lin) put(lin);
            if (y) put(1);
        }
        const int nq = (int) (r2q >> 16), c1sel = (int) (tabs >> 24);
        for (int q = 0; q < nq; q++) {
            const unsigned short *v = ix + 2 * npairs + 4 * q;
            const int code = (v[0] << 3) + (v[1] << 2) + (v[2] << 1) + v[3];
            put(c1sel == 1 ? 4 : L.quada_len[code & 15]);
            for (int k = 0; k < 4; k++) if (v[k]) put(1);
        }
    }
}

// frames_per_stream = frames a stream produces per call (nframes, or 2 nframes at the MPEG-2 rates where every
// granule is a frame); lsf selects that layout.  The grid is sized to fill the chip once; a workgroup stages the
// code tables once and then takes every gridDim.x-th frame.
__global__ __launch_bounds__(256) void k_pack(const HxStream *__restrict__ st, const HxParams *__restrict__ prm,
                                              const HxGlobalTabs *__restrict__ gt, const short *__restrict__ ixq,
                                              const unsigned *__restrict__ sgn, const HxSegOut *__restrict__ seg,
                                              const HxFrameOut *__restrict__ frm, const HxSlot *__restrict__ slots,
                                              unsigned char *__restrict__ out, long long out_stride, unsigned char *__restrict__ packet,
                                              int *__restrict__ status, int frames_per_stream, int NG, int lsf, long long nframes_total,
                                              int solo, HxStream *__restrict__ st_w, const int *__restrict__ pre_len, const int *__restrict__ out_bytes,
                                              const int *__restrict__ carry_len, unsigned *__restrict__ frames_out,
                                              unsigned char *__restrict__ host_out, const int *__restrict__ seq_src)
{
    __shared__ PackLds L;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform, made provably so)
    for (int i = tid; i < 1408; i += 256) { L.huff_code[i] = gt->huff_code[i]; L.huff_len[i] = gt->huff_len[i]; }
    if (tid < 32) {
        const int dim = (tid >= 16) ? 16 : gt->huff_dim[tid], lin = (tid >= 16) ? gt->huff_lin[tid] : 0;
        L.tabpk[tid] = gt->huff_off[tid] | (dim << 12) | (lin << 20);
    }
    if (tid < 16) { L.quada_code[tid] = gt->quada_code[tid]; L.quada_len[tid] = gt->quada_len[tid]; }
    // solo > 0 (a handful of frames in all - the one-stream encoder's calls: hx_cabi.hip launches ONE workgroup): this
    // workgroup also does what k_pack_pre and k_pack_carry do for the call's `solo` streams - the pending frames' images to the
    // head of `out` before the packing, the incomplete ones' back into the stream state behind it - two launches less per call.
    if (solo > 0) {
        for (int s = 0; s < solo; s++) {
            unsigned char *dst = out + (long long) s * out_stride;
            const int n = pre_len[s];
            for (int i = tid; i < n; i += 256) dst[i] = st_w[s].main_buf[i];      // (st_w: the pointer the images go back through below)
        }
        __syncthreads();
    }
    for (long long fr = blockIdx.x; fr < nframes_total; fr += gridDim.x) {
        const int s = (int) (fr / frames_per_stream), f = (int) (fr % frames_per_stream);
        for (int i = tid; i < 640; i += 256) L.bitw[i] = 0;
        if (tid == 0) L.negflag = 0;
        const HxParams *p = prm + __builtin_amdgcn_readfirstlane(st[s].cls);
        const int nchan = p->nchan, hdr = 4 + p->side_bytes;
        // wave w <-> segment (granule, channel)
        const int g = lsf ? f : 2 * f + (w >> 1), ch = w & 1;
        const bool mine = (lsf ? w < 2 : true) && ch < nchan;
        const long long unit = ((long long) s * NG + g) * 2 + ch;
        const HxSegOut *so = seg + unit;
        // everything the wave needs of its segment in one round trip: the record's six words, its scalefactor
        // field, the lines and the signs (requested whether or not the segment turns out to be empty)
        int4 h0 = make_int4(0, 0, 0, 0);
        int2 h1 = make_int2(0, 0);
        int fld = 0;
        const HxFrameOut fo = frm[fr];
        const HxSlot *s4 = fo.near;
        if (mine) {
            h0 = *reinterpret_cast<const int4 *>(so);
            h1 = reinterpret_cast<const int2 *>(so)[2];
            if (lane < 40) fld = so->sf[lane];
            const uint4 *sx = reinterpret_cast<const uint4 *>(ixq + unit * 576);
            const uint4 *ss = reinterpret_cast<const uint4 *>(sgn + unit * HX_SGN_WORDS);
            uint4 *dx = reinterpret_cast<uint4 *>(&L.ix[w][0]), *ds = reinterpret_cast<uint4 *>(&L.sg[w][0]);
            const uint4 a0 = sx[lane], a1 = sx[64 + (lane & 7)], a2 = ss[min(lane, HX_SGN_WORDS / 4 - 1)];
            dx[lane] = a0;
            if (lane < 8) dx[64 + lane] = a1;
            if (lane < HX_SGN_WORDS / 4) ds[lane] = a2;
        }
        const int not_null = h1.y;
        if (lane < 40) L.sf[w][lane] = (unsigned short) fld;
        if (lane == 0) { int *h = L.seghdr[w]; h[0] = h0.x; h[1] = h0.y; h[2] = h0.z; h[3] = h0.w; h[4] = h1.x; h[5] = h1.y; h[7] = mine ? 1 : 0; }
        __syncthreads();
        if (mine) {
            int pos = h0.x;                                 // start_bit
            {   // scalefactor fields in transmission order: value | length << 8 | negative << 15
                const int len = (fld >> 8) & 15;
                const int incl = hx_wave_scan(len);
                if (len) put_bits(L, pos + incl - len, (unsigned) (fld & 255) & ((1u << len) - 1u), len);
                pos += __builtin_amdgcn_readlane(incl, 63);
                if (__any(fld & 0x8000) && lane == 0) L.negflag = 1;
            }
            if (not_null) {
                const int hb = pos;
                pos = pack_huff(L, pos, (unsigned) h0.z, (unsigned) h0.w, (unsigned) h1.x, L.ix[w], L.sg[w], lane);
                if (pos - hb != h0.y && lane == 0) atomicOr(status, 4);     // counted and packed Huffman bits must agree
            }
        }
        __syncthreads();
        if (L.negflag) {        // (workgroup-uniform)
            if (tid == 0) unmasked_writer_fixup(L, lsf ? 2 : 4);
            __syncthreads();
        }
        // the frame's main data (zero stuffing up to byte_min included) into the pending slots, oldest first; the first
        // four slots' offsets and sizes are in registers (requested with the segment data), the walk rarely goes further
        const HxSlot *sl = slots + (long long) s * (frames_per_stream + HX_SLOTS_EXTRA);
        unsigned char *o = out + (long long) s * out_stride;
        for (int i = tid; i < fo.bytes; i += 256) {
            const unsigned char v = (i < fo.raw_bytes) ? (unsigned char) (L.bitw[i >> 2] >> (24 - 8 * (i & 3))) : 0;
            int q = fo.main_bytes + i, off;
            if (q < s4[0].mf) off = s4[0].off;
            else if ((q -= s4[0].mf) < s4[1].mf) off = s4[1].off;
            else if ((q -= s4[1].mf) < s4[2].mf) off = s4[2].off;
            else if ((q -= s4[2].mf) < s4[3].mf) off = s4[3].off;
            else {
                q -= s4[3].mf;
                int k = fo.first_slot + 4, cap = sl[k].mf;
                while (q >= cap) { q -= cap; k++; cap = sl[k].mf; }
                off = sl[k].off;
            }
            o[off + hdr + q] = v;
        }
        if (fo.packet_off >= 0)     // *_Packet outputs: the unpadded main data behind the packet's own header and side info
            for (int i = tid; i < fo.raw_bytes; i += 256) packet[fo.packet_off + i] = (unsigned char) (L.bitw[i >> 2] >> (24 - 8 * (i & 3)));
        __syncthreads();            // the bit buffer is cleared for the next frame
    }
    if (solo > 0) {
        // (every byte of `out` this workgroup wrote is visible to it behind the barrier above: one CU, one L1)
        for (int s = 0; s < solo; s++) {
            HxStream *ss = st_w + s;
            if (frames_out && tid == 0) frames_out[s] = ss->tot_frames_out;
            const unsigned char *src = out + (long long) s * out_stride + out_bytes[s];
            const int n = carry_len[s];
            for (int i = tid; i < n; i += 256) ss->main_buf[i] = src[i];
        }
        // The one-stream encoder's graph (hx_cabi.hip, hx_enc): the call's results go straight to page-locked host memory -
        // [byte count | frame counter | sequence word | ... 256 | bitstream] - and the sequence word last, behind system-scope
        // fences: when the host sees it change, the rest has landed.  (Round 6 first had copy nodes bring the results down and
        // polled the last of them: one call in 20 000 read the bytes before the copy ahead of it had landed - copies of one
        // stream complete in order on the device, their writes do not arrive in order in host memory.)
        if (host_out) {
            const int nb = out_bytes[0];
            const uint4 *src16 = reinterpret_cast<const uint4 *>(out);
            uint4 *dst16 = reinterpret_cast<uint4 *>(host_out + 256);
            for (int i = tid; i < (nb + 15) / 16; i += 256) dst16[i] = src16[i];
            __threadfence_system();
            __syncthreads();
            if (tid == 0) {
                reinterpret_cast<int *>(host_out)[0] = nb;
                reinterpret_cast<unsigned *>(host_out)[1] = st_w[0].tot_frames_out;
                __threadfence_system();
                __hip_atomic_store(reinterpret_cast<int *>(host_out) + 2, *seq_src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Frames whose slot is not full yet travel to the next call in the stream state: their images (headers, side
// information, the main data written so far) follow the complete frames in `out` (k_pack_carry, after the packing),
// and are put at the head of the next call's `out` before its packing (k_pack_pre).  The lengths are per call
// (written by that call's allocator launch), because the next call's allocator may already have run.
__global__ __launch_bounds__(64) void k_pack_carry(HxStream *__restrict__ st, const unsigned char *__restrict__ out, long long out_stride,
                                                   const int *__restrict__ out_bytes, const int *__restrict__ carry_len, unsigned *__restrict__ frames_out)
{
    const int s = blockIdx.x;
    HxStream *ss = st + s;
    // (the one-stream encoder's graph copies the stream's frame counter down with the call's byte count: one copy instead of two)
    if (frames_out && threadIdx.x == 0) frames_out[s] = ss->tot_frames_out;
    const unsigned char *src = out + (long long) s * out_stride + out_bytes[s];
    const int n = carry_len[s];
    for (int i = threadIdx.x; i < n; i += 64) ss->main_buf[i] = src[i];
}
__global__ __launch_bounds__(64) void k_pack_pre(const HxStream *__restrict__ st, unsigned char *__restrict__ out, long long out_stride,
                                                 const int *__restrict__ pre_len)
{
    const int s = blockIdx.x;
    const HxStream *ss = st + s;
    unsigned char *dst = out + (long long) s * out_stride;
    const int n = pre_len[s];
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = ss->main_buf[i];
}

// Workgroup order of the next allocator launch: streams by this launch's duration, longest first (a counting sort on
// 1024 duration classes; the order within a class does not matter).  One workgroup.
__global__ __launch_bounds__(1024) void k_order(const unsigned *__restrict__ dur, int *__restrict__ order, int S)
{
    __shared__ unsigned hist[1024];
    __shared__ unsigned lo, hi;
    const int tid = threadIdx.x;
    if (tid == 0) { lo = 0xFFFFFFFFu; hi = 0; }
    hist[tid] = 0;
    __syncthreads();
    unsigned mn = 0xFFFFFFFFu, mx = 0;
    for (int i = tid; i < S; i += 1024) { const unsigned d = dur[i]; mn = min(mn, d); mx = max(mx, d); }
    atomicMin(&lo, mn);
    atomicMax(&hi, mx);
    __syncthreads();
    const unsigned base = lo;
    const unsigned long long range = (unsigned long long) (hi - lo) + 1;
    for (int i = tid; i < S; i += 1024) atomicAdd(&hist[1023 - (unsigned) (((unsigned long long) (dur[i] - base) * 1024) / range)], 1u);
    __syncthreads();
    if (tid == 0) { unsigned acc = 0; for (int b = 0; b < 1024; b++) { const unsigned c = hist[b]; hist[b] = acc; acc += c; } }
    __syncthreads();
    for (int i = tid; i < S; i += 1024) order[atomicAdd(&hist[1023 - (unsigned) (((unsigned long long) (dur[i] - base) * 1024) / range)], 1u)] = i;
}
