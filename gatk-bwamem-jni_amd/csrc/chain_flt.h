// chain_flt.h -- the overlap loop of mem_chain_flt (upstream bwamem.c) in its two forms: one lane per read,
// and the whole wavefront for a read with many chains.  Included by k_chain.hip (and by the unit wrappers in tests/gpu_units).
#pragma once
#include "dev_common.h"
#include "wave_ops.h"

DEV int4 kept_entry(const Chain& c, const Seed* seeds)
{   // {query begin, query end, weight | is_alt << 31, first shadowed chain}
    int4 e; e.x = seeds[c.seed0].qbeg; e.y = seeds[c.last].qbeg + seeds[c.last].len;
    e.z = (int)((c.w & 0x7fffffffu) | (c.is_alt ? 0x80000000u : 0u)); e.w = -1;
    return e;
}
// mem_chain_flt's test of chain i = (bi, ei, wi, i_alt) against the kept chain kj: does it overlap significantly, and does
// upstream's loop stop there (i is dropped)?
DEV void flt_test(const MemOpt& opt, int bi, int ei, int wi, bool i_alt, const int4& kj, bool& ovl, bool& brk)
{
    const int bj = kj.x, ej = kj.y, wj = kj.z & 0x7fffffff;
    const bool j_alt = kj.z < 0;
    const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
    ovl = brk = false;
    if (e_min > b_max && (!j_alt || i_alt)) {
        const int li = ei - bi, lj = ej - bj;
        const int min_l = li < lj ? li : lj;
        if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
            ovl = true;
            brk = (float)wi < (float)wj * opt.drop_ratio && wj - wi >= opt.min_seed_len << 1;
        }
    }
}

// mem_chain_flt's overlap loop, one lane: every chain against the chains kept so far.  Returns n_kept.
DEV int chain_flt_lane(const MemOpt& opt, const Seed* seeds, Chain* a, int n_chn, int4* kept)
{
    // The kept chains, packed in 16 bytes each.  A read in a repeat family has hundreds of chains, every one of them kept, so
    // this loop runs n^2 / 2 times for it: one independent 16-byte load per step instead of a chain -> first seed / last seed
    // chase through three dependent loads.
    int n_kept = 0;
    a[0].kept = 3;
    kept[n_kept++] = kept_entry(a[0], seeds);
    for (int i = 1; i < n_chn; ++i) {
        int large_ovlp = 0, k;
        const Chain ci = a[i];
        const int4 ei = kept_entry(ci, seeds);
        for (k = 0; k < n_kept; ++k) {
            const int4 kj = kept[k];
            bool ovl, brk;
            flt_test(opt, ei.x, ei.y, (int)ci.w, ci.is_alt != 0, kj, ovl, brk);
            if (ovl) {
                large_ovlp = 1;
                if (kj.w < 0) ((int32_t*)&kept[k])[3] = i;
                if (brk) break;
            }
        }
        if (k == n_kept) {
            kept[n_kept++] = ei;
            a[i].kept = large_ovlp ? 2 : 3;
        }
    }
    return n_kept;
}

// The same loop for a read with many chains, by the whole wavefront: 64 kept chains per step.  Upstream walks the kept chains
// in order and stops at the first one that drops chain i; every significantly overlapping kept chain up to and including
// that one is marked as shadowing i if it shadows nothing yet.  That is a ballot (where does the walk stop?) and per-lane
// marks below the stop.  Lane 0 appends; the barriers make its stores visible to the lanes that load them next.
// Called by all 64 lanes of the workgroup with uniform arguments.
DEV int chain_flt_wave(const MemOpt& opt, const Seed* seeds, Chain* a, int n_chn, int4* kept)
{
    const int lane = threadIdx.x & 63;
    int n_kept = 1;
    if (lane == 0) { a[0].kept = 3; kept[0] = kept_entry(a[0], seeds); }
    __syncthreads();
    for (int i0 = 0; i0 < n_chn; i0 += 64) {
        int4 mine; mine.x = mine.y = mine.z = 0; mine.w = -1;
        if (i0 + lane < n_chn) mine = kept_entry(a[i0 + lane], seeds);          // the next 64 chains' summaries, one per lane
        const int jn = n_chn - i0 < 64 ? n_chn - i0 : 64;
        for (int j = i0 == 0 ? 1 : 0; j < jn; ++j) {
            const int i = i0 + j;
            int4 ei; ei.x = wave_readlane(mine.x, j); ei.y = wave_readlane(mine.y, j); ei.z = wave_readlane(mine.z, j); ei.w = -1;
            const int wi = ei.z & 0x7fffffff;
            const bool i_alt = ei.z < 0;
            bool large_ovlp = false, dropped = false;
            for (int k0 = 0; k0 < n_kept && !dropped; k0 += 64) {
                const int k = k0 + lane;
                bool ovl = false, brk = false;
                int4 kj; kj.w = 0;
                if (k < n_kept) { kj = kept[k]; flt_test(opt, ei.x, ei.y, wi, i_alt, kj, ovl, brk); }
                const uint64_t stop = __ballot(brk);
                uint64_t marks = __ballot(ovl);
                if (stop) { marks &= (2ull << (__ffsll((long long)stop) - 1)) - 1; dropped = true; }
                if (marks) large_ovlp = true;
                if ((marks >> lane & 1) && kj.w < 0) ((int32_t*)&kept[k])[3] = i;
            }
            if (!dropped) {
                if (lane == 0) { kept[n_kept] = ei; a[i].kept = large_ovlp ? 2 : 3; }
                ++n_kept;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    return n_kept;
}
