/* TEST INFRASTRUCTURE (CPU): the frame marcher's block hop (csrc/march.hip, march_probe with block bits) restated in C next to the
 * cell-by-cell walk it replaces (raymarching.cu:357-401 in the Synthetic-NeRF setting: one cascade, constant step), float for float:
 * x = fmaf(t, d, o), everything else separate operations (compile with -ffp-contract=off).  block_hop_compare() marches every ray both
 * ways and counts the rays whose emitted samples differ in any bit; tests/test_block_hop_proto_cpu.py drives it over random and
 * adversarial rays, pins walk() to the oracle's raymarching_test, and shows that WITHOUT the slack rule rays do differ. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }

typedef struct { const uint8_t* bitfield; const uint32_t* block_any; float dt, bound, bound_inv; int G; } Scene;

/* one probe at t: 1 = occupied (sample here), 0 = empty and *t_next is where the walk continues.  use_blocks: the hop rule. */
static int check_chain = 1;
static int probe(const Scene* sc, const float* o, const float* d, const float* di, float t, int use_blocks, float hop_slack,
                 float* t_next, long long* stats) {
    const float x = fmaf(t, d[0], o[0]), y = fmaf(t, d[1], o[1]), z = fmaf(t, d[2], o[2]);
    const float G = (float)sc->G, gm1 = G - 1.0f;
    const int nx = (int)fmaxf(0.0f, fminf(0.5f * (x * sc->bound_inv + 1) * G, gm1));
    const int ny = (int)fmaxf(0.0f, fminf(0.5f * (y * sc->bound_inv + 1) * G, gm1));
    const int nz = (int)fmaxf(0.0f, fminf(0.5f * (z * sc->bound_inv + 1) * G, gm1));
    const uint32_t idx = morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    int occ = 1;
    if (use_blocks) {
        occ = (sc->block_any[idx >> 14] >> ((idx >> 9) & 31)) & 1;
        if (!occ) {
            const float g8 = 8.0f / G;
            const float ux = ((((nx >> 3) + 0.5f + 0.5f * copysignf(1.0f, d[0])) * g8 * 2 - 1) * sc->bound - x) * di[0];
            const float uy = ((((ny >> 3) + 0.5f + 0.5f * copysignf(1.0f, d[1])) * g8 * 2 - 1) * sc->bound - y) * di[1];
            const float uz = ((((nz >> 3) + 0.5f + 0.5f * copysignf(1.0f, d[2])) * g8 * 2 - 1) * sc->bound - z) * di[2];
            const float tau = t + fmaxf(0.0f, fminf(ux, fminf(uy, uz)));
            uint32_t tb, ub, db;
            memcpy(&tb, &t, 4); memcpy(&ub, &tau, 4); memcpy(&db, &sc->dt, 4);
            const int sh = (int)(tb >> 23) - (int)(db >> 23);
            if (tau - t < 0.25f && (ub >> 23) == (tb >> 23) && sh >= 1 && sh <= 23 && (db >> 23) != 0u && (tb >> 23) != 0u) {
                /* the landing point and its predecessor in closed form (one binade: the step adds the same number of ulps every time) */
                const uint32_t md = (db & 0x7fffffu) | 0x800000u;
                const uint32_t rem = md & ((1u << sh) - 1u), half = 1u << (sh - 1);
                const uint32_t delta = (md >> sh) + (rem > half ? 1u : 0u);
                const uint32_t diff = (ub & 0x7fffffu) - (tb & 0x7fffffu);
                uint32_t k = delta ? (uint32_t)((float)diff / (float)delta) : 0u;       /* (the kernel multiplies by v_rcp: within one as well) */
                if (use_blocks == 2 && k > 0) k -= 1;                                    /* exercise the upward correction */
                if (use_blocks == 3) k += 1;                                             /* ... and the downward one */
                if (k * delta < diff) ++k;
                if (k * delta < diff) ++k;
                if (k > 1u && (k - 1u) * delta >= diff) --k;
                if (k == 0u) k = 1u;
                const uint32_t mk = (tb & 0x7fffffu) + k * delta;
                if (rem != half && delta != 0u && mk < 0x800000u) {
                    const uint32_t qb = (tb & 0xff800000u) | mk, pb = (tb & 0xff800000u) | (mk - delta);
                    float tt, prev;
                    memcpy(&tt, &qb, 4); memcpy(&prev, &pb, 4);
                    if (check_chain) {                       /* the closed form IS the chain of adds */
                        float c = t, cp = t;
                        do { cp = c; c += sc->dt; } while (c < tau);
                        if (c != tt || cp != prev) stats[4]++;
                    }
                    if (tt - tau > hop_slack && tau - prev > hop_slack) { *t_next = tt; stats[0]++; return 0; }
                }
                stats[1]++;
            }
        }
    }
    if (occ) occ = (sc->bitfield[idx >> 3] >> (idx & 7)) & 1;
    if (!occ) {
        const float ginv = 1.0f / G;
        const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, d[0])) * ginv * 2 - 1) * sc->bound - x) * di[0];
        const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, d[1])) * ginv * 2 - 1) * sc->bound - y) * di[1];
        const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, d[2])) * ginv * 2 - 1) * sc->bound - z) * di[2];
        const float t_target = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        float tt = t;
        if (t_target + sc->dt == t_target) tt = INFINITY;
        else do { tt += sc->dt; } while (tt < t_target);
        *t_next = tt; stats[2]++;
    }
    return occ;
}

/* emitted t of one ray, at most cap of them; returns how many */
static int walk(const Scene* sc, const float* o, const float* d, float t1, float t2, int use_blocks, float slack_scale, float* ts, int cap,
                long long* stats) {
    const float di[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float hop_slack = slack_scale * 8.0f * (fabsf(t2) * 1.2e-7f + 1.2e-7f + 1.2e-7f * fmaxf(fabsf(di[0]), fmaxf(fabsf(di[1]), fabsf(di[2]))));
    float t = t1;
    int s = 0, iters = 0;
    while (t < t2 && s < cap && ++iters < (1 << 20)) {
        float t_next;
        if (probe(sc, o, d, di, t, use_blocks, hop_slack, &t_next, stats)) { ts[s++] = t; t += sc->dt; }
        else t = t_next;
    }
    return s;
}

void block_hop_block_bits(const uint8_t* bitfield, uint32_t* block_any) {        /* 128^3 grid: 4096 blocks of 512 Morton-consecutive cells */
    memset(block_any, 0, 128 * sizeof(uint32_t));
    for (int b = 0; b < 4096; ++b) {
        int any = 0;
        for (int i = 0; i < 64; ++i) any |= bitfield[64 * b + i];
        if (any) block_any[b >> 5] |= 1u << (b & 31);
    }
}

/* rays whose samples differ between the cell-by-cell walk and the block-hop walk.  stats[0..2]: block hops taken, block hops declined,
 * cell hops; stats[3]: samples emitted; stats[4]: closed-form landings that are not the chain of adds' (must stay 0); first_bad: index of the first differing ray or -1. */
long long block_hop_compare(const uint8_t* bitfield, const float* rays_o, const float* rays_d, const float* hits, long long n_rays,
                            float scale, int max_samples, float slack_scale, long long* stats, long long* first_bad) {
    uint32_t block_any[128];
    block_hop_block_bits(bitfield, block_any);
    Scene sc = {bitfield, block_any, 1.7320508075688772f / max_samples, fminf(0.5f, scale), 0.0f, 128};
    sc.bound_inv = 1 / sc.bound;
    long long bad = 0, dummy[5] = {0, 0, 0, 0, 0};
    *first_bad = -1;
    static float a[4096], b[4096];
    for (long long r = 0; r < n_rays; ++r) {
        const float t1 = hits[2 * r], t2 = hits[2 * r + 1];
        const int na = walk(&sc, rays_o + 3 * r, rays_d + 3 * r, t1, t2, 0, 0.0f, a, 4096, dummy);
        const int nb = walk(&sc, rays_o + 3 * r, rays_d + 3 * r, t1, t2, 1 + (int)(r % 3), slack_scale, b, 4096, stats);
        stats[3] += na;
        if (na != nb || memcmp(a, b, sizeof(float) * (size_t)na) != 0) { if (*first_bad < 0) *first_bad = r; ++bad; }
    }
    return bad;
}

/* the cell-by-cell walk alone, row-major (n_rays, cap) with counts: what the test pins against the oracle's raymarching_test */
void block_hop_walk(const uint8_t* bitfield, const float* rays_o, const float* rays_d, const float* hits, long long n_rays, float scale,
                    int max_samples, int cap, float* ts, int32_t* counts) {
    Scene sc = {bitfield, 0, 1.7320508075688772f / max_samples, fminf(0.5f, scale), 0.0f, 128};
    sc.bound_inv = 1 / sc.bound;
    long long dummy[5] = {0, 0, 0, 0, 0};
    for (long long r = 0; r < n_rays; ++r)
        counts[r] = walk(&sc, rays_o + 3 * r, rays_d + 3 * r, hits[2 * r], hits[2 * r + 1], 0, 0.0f, ts + (size_t)r * cap, cap, dummy);
}
