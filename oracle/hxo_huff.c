/* hxo_huff.c - ORACLE (test infrastructure): Huffman region split / table choice / bit count
 * and bitstream packing.  Restates bitalloc.cpp:310-811, cnt.c:96-325, l3pack.c:107-1187.
 * All-integer: must be bit-exact. */
#include <string.h>
#include <stdlib.h>
#include "hxo_int.h"

/* ---- candidate Huffman tables by region maximum (cnttab.h:38-62, bitalloc.cpp:310-420) ---- */
typedef struct { int ncand; int t[4]; int tmax; } cand_t;

static void candidates(int rmax, cand_t *c)
{
    static const signed char small[23][5] = {
        {0, 0, 0, 0, 0}, {1, 3, 0, 0, 1}, {2, 3, 0, 0, 2}, {5, 6, 0, 0, 3}, {7, 8, 9, 12, 5}, {7, 8, 9, 12, 5},
        {10, 11, 12, 15, 7}, {10, 11, 12, 15, 7}, {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15},
        {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15}, {13, 15, 0, 0, 15},
        {16, 24, 0, 0, 16}, {17, 24, 0, 0, 18}, {17, 24, 0, 0, 18}, {18, 24, 0, 0, 22}, {18, 24, 0, 0, 22},
        {18, 24, 0, 0, 22}, {18, 24, 0, 0, 22}};
    static const short big[9][3] = {
        {30, 19, 24}, {46, 25, 20}, {78, 20, 26}, {142, 27, 21}, {270, 21, 28},
        {526, 29, 22}, {1038, 22, 30}, {2062, 30, 23}, {8206, 31, 23}};
    int i;
    c->t[2] = c->t[3] = 0;
    if (rmax <= 22) {
        for (i = 0; i < 4; i++) c->t[i] = small[rmax][i];
        c->tmax = small[rmax][4];
        c->ncand = (rmax == 0) ? 0 : ((rmax >= 4 && rmax <= 7) ? 4 : 2);
        return;
    }
    for (i = 0; i < 8; i++) if (rmax <= big[i][0]) break;
    c->t[0] = big[i][1];
    c->t[1] = big[i][2];
    c->tmax = big[i][0];
    c->ncand = 2;
}

/* coded length of one pair in table t: Huffman length + sign bits + linbits */
static int pair_len(int t, int x, int y)
{
    int lin = hxo_huff_linbits(t), cx = x > 15 ? 15 : x, cy = y > 15 ? 15 : y;
    int n = hxo_huff_len(t, (t >= 16) ? cx : x, (t >= 16) ? cy : y);
    if (x) n++;
    if (y) n++;
    if (t >= 16) { if (x >= 15) n += lin; if (y >= 15) n += lin; }
    return n;
}

/* shared with the short-block counter (hxo_short.c) */
void hxo_huff_candidates(int rmax, int *ncand, int t[4], int *tmax)
{
    cand_t c;
    candidates(rmax, &c);
    *ncand = c.ncand; *tmax = c.tmax;
    t[0] = c.t[0]; t[1] = c.t[1]; t[2] = c.t[2]; t[3] = c.t[3];
}
int hxo_huff_pair_len(int t, int x, int y) { return pair_len(t, x, y); }

/* cnt.c:96-288: sum the candidates' lengths over a region, pick the shortest; ties go to the
   higher candidate index.  Returns bits, *index = chosen candidate. */
static int count_region(const cand_t *c, const int *ix, int n, int *index)
{
    int b[4] = {0, 0, 0, 0}, i, k, bits;
    *index = 0;
    if (c->ncand == 0 || n <= 0) return 0;
    for (i = 0; i < n; i += 2)
        for (k = 0; k < c->ncand; k++) b[k] += pair_len(c->t[k], ix[i], ix[i + 1]);
    for (k = 0; k < 4; k++) b[k] &= 0xFFFF;
    if (b[0] < b[1]) { bits = b[0]; *index = 0; } else { bits = b[1]; *index = 1; }
    if (c->ncand == 4) {
        if (b[2] <= bits) { bits = b[2]; *index = 2; }
        if (b[3] <= bits) { bits = b[3]; *index = 3; }
    }
    return bits;
}

/* cnt.c:292-325 */
static int count_quads(const int *ix, int nquads, int *index)
{
    int i, a = 0, b = 0;
    *index = 0;
    if (nquads <= 0) return 0;
    for (i = 0; i < nquads; i++, ix += 4) {
        int v = (ix[0] << 3) + (ix[1] << 2) + (ix[2] << 1) + ix[3];
        int pop = ix[0] + ix[1] + ix[2] + ix[3];
        a += hxo_quada_len(v & 15) + pop;
        b += 4 + pop;
    }
    if (a < b) { *index = 0; return a; }
    *index = 1;
    return b;
}

static int region_max(const int *ixmax, int a, int b)
{
    int i, m = 0;
    for (i = a; i < b; i++) if (m < ixmax[i]) m = ixmax[i];
    return m;
}

/* bitalloc.cpp:470-633 (block type 0) and :637-756 (block types 1/3, fixed region0 = 8 sfb) */
int hxo_count_bits(const hxo_params *p, const int *ixmax, const int *ix, int ncb, int opti,
                   int block_type, hxo_huffsel *out)
{
    static const unsigned char reg0[24] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 7};
    static const unsigned char reg1[24] = {1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5, 5, 5, 5, 6, 6, 7, 7, 7, 8, 8, 8, 8};
    const int *sb = p->startBand_l, *nb = p->nBand_l;
    int cb[4], i, j, n, n0, nbig, nquads, bits, idx;
    cand_t c[3];

    for (i = ncb - 1; i >= 0; i--) if (ixmax[i] > 0) break;
    cb[3] = i + 1;
    for (; i >= 0; i--) if (ixmax[i] > 1) break;
    cb[2] = i + 1;
    if (block_type == 0) {
        if (cb[2] < 2) { cb[2] = 2; if (cb[3] < cb[2]) cb[3] = cb[2]; }
    } else {
        cb[0] = 8;
        cb[2] = HXO_MAX(cb[2], 8);
        cb[3] = HXO_MAX(cb[3], cb[2]);
        cb[1] = cb[0];
    }
    j = sb[cb[2]];
    n = nb[cb[2] - 1];
    for (i = 0; i < n; i++) { j--; if (ix[j] > 1) break; }
    nbig = (j + 2) & (~1);
    if (block_type == 0) { if (nbig < sb[2]) nbig = sb[2]; }
    else { if (nbig < sb[8]) nbig = sb[8]; }
    j = sb[cb[3]];
    n = nb[cb[3] - 1];
    for (i = 0; i < n; i++) { j--; if (ix[j] > 0) break; }
    nquads = (j + 4 - nbig) >> 2;
    if (block_type != 0) nquads = HXO_MAX(nquads, 0);

    if (block_type == 0) {
        cb[0] = reg0[cb[2]];
        cb[1] = reg0[cb[2]] + reg1[cb[2]];
        if (cb[0] < 1) cb[0] = 1;
        if (cb[1] <= cb[0]) cb[1] = cb[0] + 1;
        if (cb[1] > cb[0] + 8) cb[1] = cb[0] + 8;
        candidates(region_max(ixmax, 0, cb[0]), &c[0]);
        candidates(region_max(ixmax, cb[0], cb[1]), &c[1]);
        candidates(region_max(ixmax, cb[1], cb[2]), &c[2]);
        if (opti & 1) {
            if (c[2].tmax < c[1].tmax) {            /* shrink region 1, grow region 2 */
                for (j = cb[1] - 1; j > cb[0]; j--) if (ixmax[j] > c[2].tmax) break;
                cb[1] = j + 1;
            }
            if (c[1].tmax < c[0].tmax) {            /* shrink region 0, grow region 1 (<= 8 bands) */
                n = cb[1] - 8;
                if (n < 1) n = 1;
                for (j = cb[0] - 1; j > n; j--) if (ixmax[j] > c[1].tmax) break;
                cb[0] = j + 1;
            }
        }
        n0 = sb[cb[0]];
        n = sb[cb[1]];
        bits = count_region(&c[0], ix, n0, &idx);
        out->table[0] = c[0].t[idx];
        bits += count_region(&c[1], ix + n0, n - n0, &idx);
        out->table[1] = c[1].t[idx];
        bits += count_region(&c[2], ix + n, nbig - n, &idx);
        out->table[2] = c[2].t[idx];
    } else {
        candidates(region_max(ixmax, 0, cb[0]), &c[0]);
        candidates(0, &c[1]);
        candidates(region_max(ixmax, cb[0], cb[2]), &c[2]);
        n0 = sb[cb[0]];
        n = sb[cb[1]];
        bits = count_region(&c[0], ix, n0, &idx);
        out->table[0] = c[0].t[idx];
        bits += count_region(&c[2], ix + n, nbig - n, &idx);
        out->table[2] = c[2].t[idx];
        out->table[1] = out->table[2];
    }
    bits += count_quads(ix + nbig, nquads, &idx);
    out->table[3] = idx;
    out->cbreg[0] = cb[0]; out->cbreg[1] = cb[1]; out->cbreg[2] = cb[2];
    out->nbig = nbig;
    out->nquads = nquads;
    out->bits = bits;
    return bits;
}

/* bitalloc.cpp:758-811 */
void hxo_huffsel_to_gr(const hxo_params *p, const hxo_huffsel *s, hxo_gr *g)
{
    int n0, n1, n2;
    if (s->bits <= 0) {
        g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
        g->big_values = g->region0_count = g->region1_count = 0;
        g->aux_nreg[0] = g->aux_nreg[1] = g->aux_nreg[2] = 0;
        g->aux_nquads = 0;
        g->count1table_select = 0;
        return;
    }
    g->table_select[0] = s->table[0];
    g->table_select[1] = s->table[1];
    g->table_select[2] = s->table[2];
    g->count1table_select = s->table[3];
    g->big_values = s->nbig >> 1;
    g->region0_count = s->cbreg[0] - 1;
    g->region1_count = (s->cbreg[1] - s->cbreg[0]) - 1;
    g->region1_count = HXO_MAX(g->region1_count, 0);
    n0 = p->startBand_l[s->cbreg[0]];
    n1 = p->startBand_l[s->cbreg[1]];
    n2 = p->startBand_l[s->cbreg[2]];
    if (n2 > s->nbig) n2 = s->nbig;
    if (n1 > n2) n1 = n2;
    if (n0 > n1) n0 = n1;
    n2 = n2 - n1;
    n1 = n1 - n0;
    g->aux_nreg[0] = n0 >> 1;
    g->aux_nreg[1] = n1 >> 1;
    g->aux_nreg[2] = n2 >> 1;
    g->aux_nquads = s->nquads;
}

/* ---- bit writer (l3pack.c:99-152): 32-bit accumulator, bytes drained when room < need ---- */
void hxo_bw_init(hxo_bitw *w, unsigned char *out)
{
    w->buf0 = w->buf = out;
    w->bitbuf = 0;
    w->room = 32;
    w->bit_pos_start = 0;
}

void hxo_bw_put(hxo_bitw *w, unsigned x, int n)
{
    if (w->room < n)
        while (w->room < 24) {
            *w->buf++ = (unsigned char) ((unsigned) w->bitbuf >> (24 - w->room));
            w->room += 8;
        }
    w->bitbuf = (int) (((unsigned) w->bitbuf << n) | x);
    w->room -= n;
}

int hxo_bw_flush(hxo_bitw *w)
{
    while (w->room < 24) {
        *w->buf++ = (unsigned char) ((unsigned) w->bitbuf >> (24 - w->room));
        w->room += 8;
    }
    if (w->room < 32) *w->buf++ = (unsigned char) ((unsigned) w->bitbuf << (w->room - 24));
    w->room = 32;
    return (int) (w->buf - w->buf0);
}

static int bw_pos(const hxo_bitw *w) { return (int) ((w->buf - w->buf0) << 3) + (32 - w->room); }

/* slen1/slen2 from the maxima, via scalefac_compress (l3pack.c:57-73,188-212) */
static int sf_compress(int sfmax1, int sfmax2, int *slen1, int *slen2)
{
    static const unsigned char comp[5][4] = {
        {0, 1, 2, 3}, {5, 5, 6, 7}, {8, 8, 9, 10}, {4, 11, 12, 13}, {14, 14, 14, 15}};
    static const unsigned char slen[16][2] = {
        {0, 0}, {0, 1}, {0, 2}, {0, 3}, {3, 0}, {1, 1}, {1, 2}, {1, 3},
        {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2}, {3, 3}, {4, 2}, {4, 3}};
    int n, s1, s2, sc;
    n = 1; sfmax1++;
    for (s1 = 0; s1 < 4; s1++) { if (sfmax1 <= n) break; n += n; }
    n = 1; sfmax2++;
    for (s2 = 0; s2 < 3; s2++) { if (sfmax2 <= n) break; n += n; }
    sc = comp[s1][s2];
    *slen1 = slen[sc][0];
    *slen2 = slen[sc][1];
    return sc;
}

/* l3pack.c:157-214 (long blocks, no scfsi) */
int hxo_pack_sf_long(hxo_bitw *w, const hxo_scalefact *sf)
{
    int i, m1 = 0, m2 = 0, s1, s2, sc;
    w->bit_pos_start = bw_pos(w);
    for (i = 0; i < 11; i++) if (sf->l[i] > m1) m1 = sf->l[i];
    for (; i < 21; i++) if (sf->l[i] > m2) m2 = sf->l[i];
    sc = sf_compress(m1, m2, &s1, &s2);
    for (i = 0; i < 11; i++) hxo_bw_put(w, sf->l[i], s1);
    for (; i < 21; i++) hxo_bw_put(w, sf->l[i], s2);
    return sc;
}

/* l3pack.c:218-288 */
int hxo_pack_sf_short(hxo_bitw *w, const hxo_scalefact *sf)
{
    int i, k, m1 = 0, m2 = 0, s1, s2, sc;
    w->bit_pos_start = bw_pos(w);
    for (i = 0; i < 6; i++) for (k = 0; k < 3; k++) m1 = HXO_MAX(m1, sf->s[k][i]);
    for (; i < 12; i++) for (k = 0; k < 3; k++) m2 = HXO_MAX(m2, sf->s[k][i]);
    sc = sf_compress(m1, m2, &s1, &s2);
    for (i = 0; i < 6; i++) for (k = 0; k < 3; k++) hxo_bw_put(w, sf->s[k][i], s1);
    for (; i < 12; i++) for (k = 0; k < 3; k++) hxo_bw_put(w, sf->s[k][i], s2);
    return sc;
}

/* slen for one MPEG-2 scalefactor group (l3pack.c:636-667) */
static int lsf_slen(int sfmax, int cap)
{
    int n = 1, s;
    sfmax++;
    for (s = 0; s < cap; s++) { if (sfmax <= n) break; n += n; }
    return s;
}

/* l3pack.c:561-729 (long) and :732-930 (short), the non-intensity branch: four groups of 6/5/5/5 long
   or 3/3/3/3 short bands, slen1..4 capped at 4/4/3/3, 9-bit scalefac_compress */
int hxo_pack_sf_lsf(hxo_bitw *w, const hxo_scalefact *sf, int block_type)
{
    static const int edge_l[5] = {0, 6, 11, 16, 21}, edge_s[5] = {0, 3, 6, 9, 12}, cap[4] = {4, 4, 3, 3};
    int i, k, g, m, slen[4];
    w->bit_pos_start = bw_pos(w);
    if (block_type == 2) {
        for (g = 0; g < 4; g++) {
            for (m = 0, k = 0; k < 3; k++) for (i = edge_s[g]; i < edge_s[g + 1]; i++) m = HXO_MAX(m, sf->s[k][i]);
            slen[g] = lsf_slen(m, cap[g]);
        }
        for (g = 0; g < 4; g++)
            for (i = edge_s[g]; i < edge_s[g + 1]; i++) for (k = 0; k < 3; k++) hxo_bw_put(w, sf->s[k][i], slen[g]);
    } else {
        for (g = 0; g < 4; g++) {
            for (m = 0, i = edge_l[g]; i < edge_l[g + 1]; i++) m = HXO_MAX(m, sf->l[i]);
            slen[g] = lsf_slen(m, cap[g]);
        }
        for (g = 0; g < 4; g++) for (i = edge_l[g]; i < edge_l[g + 1]; i++) hxo_bw_put(w, sf->l[i], slen[g]);
    }
    return slen[3] + (slen[2] << 2) + ((slen[1] + 5 * slen[0]) << 4);
}

/* l3pack.c:561-729, the intensity branch (long blocks): the right channel of an MPEG-2 joint-stereo frame with
   intensity coding.  Three groups of 7 bands; bands from nsf_stereo on carry intensity positions, and a
   scalefactor of 999 marks "no intensity" and is sent as the group's all-ones value (which a real position must
   therefore never take: the group's length grows by a bit if it would).  Rewrites the 999 entries of sf. */
int hxo_pack_sf_lsf_is(hxo_bitw *w, hxo_scalefact *sf, int nsf_stereo)
{
    static const int edge[4] = {0, 7, 14, 21}, cap[3] = {4, 4, 3};
    int g, i, m[3] = {0, 0, 0}, ism[3] = {-1, -1, -1}, ip[3] = {0, 0, 0}, slen[3];
    w->bit_pos_start = bw_pos(w);
    for (g = 0; g < 3; g++)
        for (i = edge[g]; i < edge[g + 1]; i++) {
            if (sf->l[i] >= 999) { ip[g] = 1; continue; }
            if (sf->l[i] > m[g]) m[g] = sf->l[i];
            if (g > 0 && i >= nsf_stereo && sf->l[i] > ism[g]) ism[g] = sf->l[i];
        }
    for (g = 0; g < 3; g++) slen[g] = lsf_slen(m[g], cap[g]);
    if (ism[1] == ((1 << slen[1]) - 1)) slen[1]++;
    if (ism[2] == ((1 << slen[2]) - 1)) slen[2]++;
    for (g = 0; g < 3; g++)
        if (ip[g]) for (i = edge[g]; i < edge[g + 1]; i++) if (sf->l[i] >= 999) sf->l[i] = (1 << slen[g]) - 1;
    for (g = 0; g < 3; g++) for (i = edge[g]; i < edge[g + 1]; i++) hxo_bw_put(w, sf->l[i], slen[g]);
    g = slen[2] + 6 * slen[1] + 36 * slen[0];
    return g + g + 1;
}

/* l3pack.c:421-558: long blocks with scfsi reuse between granule 0 and 1 */
int hxo_pack_sf_long_scfsi(hxo_bitw *w, int sf_save[21], const hxo_scalefact *sf, int igr,
                           int *pscfsi, int not_null)
{
    static const int edge[5] = {0, 6, 11, 16, 21};
    int i, g, t, scfsi = 0, sc = 0, m1 = 0, m2 = 0, s1, s2;
    if (igr == 0) {
        for (i = 0; i < 21; i++) sf_save[i] = sf->l[i];
    } else {
        for (g = 0; g < 4; g++) {
            for (t = 0, i = edge[g]; i < edge[g + 1]; i++) t |= (sf_save[i] - sf->l[i]);
            scfsi <<= 1;
            if (t == 0) scfsi |= 1;
        }
    }
    w->bit_pos_start = bw_pos(w);
    if (not_null) {
        for (g = 0; g < 4; g++) {
            if (scfsi & (8 >> g)) continue;
            for (i = edge[g]; i < edge[g + 1]; i++) {
                if (g < 2) { if (sf->l[i] > m1) m1 = sf->l[i]; }
                else { if (sf->l[i] > m2) m2 = sf->l[i]; }
            }
        }
        sc = sf_compress(m1, m2, &s1, &s2);
        for (g = 0; g < 4; g++) {
            if (scfsi & (8 >> g)) continue;
            for (i = edge[g]; i < edge[g + 1]; i++) hxo_bw_put(w, sf->l[i], g < 2 ? s1 : s2);
        }
    }
    *pscfsi = scfsi;
    return sc;
}

/* l3pack.c:946-1119: returns part2_3_length (scalefactor bits since bit_pos_start + Huffman bits) */
int hxo_pack_huff(hxo_bitw *w, const hxo_gr *g, const int *ix, const unsigned char *sign)
{
    int r, j, n, t, x, y, lin;
    for (r = 0; r < 3; r++) {
        n = g->aux_nreg[r];
        t = g->table_select[r];
        if (hxo_huff_dim(t) != 0) {
            lin = hxo_huff_linbits(t);
            for (j = 0; j < n; j++) {
                x = ix[2 * j]; y = ix[2 * j + 1];
                if (t >= 16) {
                    int cx = x > 15 ? 15 : x, cy = y > 15 ? 15 : y;
                    hxo_bw_put(w, hxo_huff_code(t, cx, cy), hxo_huff_len(t, cx, cy));
                    if (cx >= 15) hxo_bw_put(w, x - 15, lin);
                    if (cx) hxo_bw_put(w, sign[2 * j], 1);
                    if (cy >= 15) hxo_bw_put(w, y - 15, lin);
                    if (cy) hxo_bw_put(w, sign[2 * j + 1], 1);
                } else {
                    hxo_bw_put(w, hxo_huff_code(t, x, y), hxo_huff_len(t, x, y));
                    if (x) hxo_bw_put(w, sign[2 * j], 1);
                    if (y) hxo_bw_put(w, sign[2 * j + 1], 1);
                }
            }
        }
        ix += 2 * n;
        sign += 2 * n;
    }
    n = g->aux_nquads;
    for (j = 0; j < n; j++, ix += 4, sign += 4) {
        int v = (ix[0] << 3) + (ix[1] << 2) + (ix[2] << 1) + ix[3];
        if (g->count1table_select == 1) hxo_bw_put(w, v ^ 15, 4);
        else hxo_bw_put(w, hxo_quada_code(v), hxo_quada_len(v));
        if (v & 8) hxo_bw_put(w, sign[0], 1);
        if (v & 4) hxo_bw_put(w, sign[1], 1);
        if (v & 2) hxo_bw_put(w, sign[2], 1);
        if (v & 1) hxo_bw_put(w, sign[3], 1);
    }
    return bw_pos(w) - w->bit_pos_start;
}

/* l3pack.c:1123-1187 (stereo: 32 bytes; main_data_begin left 0, patched at emit time) */
void hxo_pack_side(unsigned char out[32], int mode, const int scfsi[2], hxo_gr gr[2][2], int nchan)
{
    hxo_bitw w;
    int igr, ch;
    hxo_bw_init(&w, out);
    hxo_bw_put(&w, 0, 9);
    hxo_bw_put(&w, 0, mode == 3 ? 5 : 3);       /* private bits: 5 in a mono frame (17-byte side info) */
    for (ch = 0; ch < nchan; ch++) hxo_bw_put(&w, scfsi[ch], 4);
    for (igr = 0; igr < 2; igr++)
        for (ch = 0; ch < nchan; ch++) {
            const hxo_gr *g = &gr[igr][ch];
            hxo_bw_put(&w, g->part2_3_length, 12);
            hxo_bw_put(&w, g->big_values, 9);
            hxo_bw_put(&w, g->global_gain, 8);
            hxo_bw_put(&w, g->scalefac_compress, 4);
            hxo_bw_put(&w, g->window_switching_flag, 1);
            if (g->window_switching_flag) {
                hxo_bw_put(&w, g->block_type, 2);
                hxo_bw_put(&w, g->mixed_block_flag, 1);
                hxo_bw_put(&w, g->table_select[0], 5);
                hxo_bw_put(&w, g->table_select[1], 5);
                hxo_bw_put(&w, g->subblock_gain[0], 3);
                hxo_bw_put(&w, g->subblock_gain[1], 3);
                hxo_bw_put(&w, g->subblock_gain[2], 3);
            } else {
                hxo_bw_put(&w, g->table_select[0], 5);
                hxo_bw_put(&w, g->table_select[1], 5);
                hxo_bw_put(&w, g->table_select[2], 5);
                hxo_bw_put(&w, g->region0_count, 4);
                hxo_bw_put(&w, g->region1_count, 3);
            }
            hxo_bw_put(&w, g->preflag, 1);
            hxo_bw_put(&w, g->scalefac_scale, 1);
            hxo_bw_put(&w, g->count1table_select, 1);
        }
    hxo_bw_flush(&w);
}

/* l3pack.c:1189-1246: one-granule MPEG-2 side info (stereo 17 bytes, mono 9); main_data_begin (8 bits) patched at emit */
void hxo_pack_side_lsf(unsigned char out[32], int mode, hxo_gr gr[2], int nchan)
{
    hxo_bitw w;
    int ch;
    hxo_bw_init(&w, out);
    hxo_bw_put(&w, 0, 8);
    hxo_bw_put(&w, 0, mode == 3 ? 1 : 2);
    for (ch = 0; ch < nchan; ch++) {
        const hxo_gr *g = &gr[ch];
        hxo_bw_put(&w, g->part2_3_length, 12);
        hxo_bw_put(&w, g->big_values, 9);
        hxo_bw_put(&w, g->global_gain, 8);
        hxo_bw_put(&w, g->scalefac_compress, 9);
        hxo_bw_put(&w, g->window_switching_flag, 1);
        if (g->window_switching_flag) {
            hxo_bw_put(&w, g->block_type, 2);
            hxo_bw_put(&w, g->mixed_block_flag, 1);
            hxo_bw_put(&w, g->table_select[0], 5);
            hxo_bw_put(&w, g->table_select[1], 5);
            hxo_bw_put(&w, g->subblock_gain[0], 3);
            hxo_bw_put(&w, g->subblock_gain[1], 3);
            hxo_bw_put(&w, g->subblock_gain[2], 3);
        } else {
            hxo_bw_put(&w, g->table_select[0], 5);
            hxo_bw_put(&w, g->table_select[1], 5);
            hxo_bw_put(&w, g->table_select[2], 5);
            hxo_bw_put(&w, g->region0_count, 4);
            hxo_bw_put(&w, g->region1_count, 3);
        }
        hxo_bw_put(&w, g->scalefac_scale, 1);
        hxo_bw_put(&w, g->count1table_select, 1);
    }
    hxo_bw_flush(&w);
}
