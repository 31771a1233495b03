// The CHAIN plan of the one-launch backward blocks (wn_respq.hip: the gated block; wn_encpq.hip: the autoencoder's encoder block), shared
// by the kernels and the host (wn_pq_chain_items: the CPU test of the plan).
#pragma once
#include <hip/hip_runtime.h>

// Items are the 32-column tiles j = 0 .. steps-1 of a clip (from t_base); chain (clip b, residue r) holds the items j = r (mod s), s = d / 32, walked
// from the highest j down: qn + 1 items for r < rm, qn otherwise.  Chain order = clips, then residues, then positions.
struct PqChain { int s, qn, rm, g, nchain, steps; };
struct PqCS { int b, r, pos, m; };                         // chain state: clip, residue, position from the top, items of the chain
__host__ __device__ __forceinline__ PqCS pq_cs_next(PqCS c, const PqChain& p) {
    if (c.pos + 1 < c.m) { c.pos += 1; return c; }
    c.pos = 0;
    c.r += 1;
    if (c.r == p.s) { c.r = 0; c.b += 1; }
    c.m = p.qn + (c.r < p.rm ? 1 : 0);
    return c;
}
// workgroup wg of nwg: its first item (the halo item, if it has one), the number of items it owns, whether a halo item precedes them
__host__ __device__ __forceinline__ void pq_chain_start(const PqChain& p, int wg, int nwg, PqCS& c, int& n_real, bool& halo) {
    int c0, p0;
    if (p.g > 0) {                                         // g segments per chain, the first (m mod g) one item longer
        c0 = wg / p.g;
        const int sg = wg - c0 * p.g;
        const int r = c0 % p.s;
        const int m = p.qn + (r < p.rm ? 1 : 0);
        const int base = m / p.g, ex = m - base * p.g;
        n_real = base + (sg < ex ? 1 : 0);
        p0 = sg * base + (sg < ex ? sg : ex);
    } else {                                               // whole chains per workgroup
        c0 = (int)((long)wg * p.nchain / nwg);
        const int c1 = (int)((long)(wg + 1) * p.nchain / nwg);
        const int b0 = c0 / p.s, r0 = c0 - b0 * p.s, b1 = c1 / p.s, r1 = c1 - b1 * p.s;
        n_real = (b1 * p.steps + r1 * p.qn + (r1 < p.rm ? r1 : p.rm)) - (b0 * p.steps + r0 * p.qn + (r0 < p.rm ? r0 : p.rm));
        p0 = 0;
    }
    halo = p0 > 0 && n_real > 0;
    c.b = c0 / p.s;
    c.r = c0 - c.b * p.s;
    c.m = p.qn + (c.r < p.rm ? 1 : 0);
    c.pos = p0 - (halo ? 1 : 0);
}

