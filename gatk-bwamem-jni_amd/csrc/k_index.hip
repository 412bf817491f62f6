// k_index.hip -- index construction on the device: forward strand -> suffix array of forward + reverse complement ->
// BWT with interleaved occ checkpoints + sampled SA, i.e. what index_pieces_from_sa (index_build.cpp) computes on the host,
// for references the host's prefix-doubling sort is far too slow for.
//
// Replaces upstream bwa_idx_build (bwtindex.c: BWT construction + bwt_bwtupdate_core + bwt_cal_sa) as reached from the
// reference at ...BwaMemIndex.c:42-63; byte-identical to the host builder (tests/test_index_device.py),
// which reproduces the reference's fixture files src/test/resources/ref.fa.{bwt,sa,pac,ann,amb} (tests/test_index.py).
//
// Suffix sorting is an MSD bucket sort on 32-base keys over the 2-bit packed text: the suffixes are bucketed by their first
// two bases; a bucket is radix-sorted on its next 32 bases (one 64-bit key); groups that are still tied are refined with
// further 32-mers, the tied set shrinking quickly -- on a genome almost every suffix is resolved by the first key, and what
// repeats leave behind takes a few dozen rounds on thousands, not billions, of suffixes.  The sorts, scans and selections
// are rocPRIM's device primitives (each over one bucket or one tied set: well below 2^31 items); the kernels around them
// index the text with 64 bits.  A suffix that runs past the end of the text is the smaller one (the sentinel), so among tied
// suffixes that have all run out the one starting later sorts first.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "index_io.h"

#define IDX_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

namespace {

struct Buf {
    void* p = nullptr; size_t bytes = 0;
    bool ensure(size_t n) { if (n <= bytes) return true; release(); n += n / 16 + 256; if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; return false; } bytes = n; return true; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return (T*)p; }
    ~Buf() { release(); }
};

// base i of the text (forward strand followed by its reverse complement) from the forward strand's 2-bit codes
__device__ inline uint32_t text_at(const uint8_t* fwd, uint64_t l, uint64_t i) { return i < l ? fwd[i] : 3u - fwd[2 * l - 1 - i]; }

// W[k] = bases 32k .. 32k+31 of the text, first base in the two most significant bits; zero beyond the end
__global__ void k_pack_words(const uint8_t* fwd, uint64_t l, uint64_t n_words, uint64_t* W)
{
    const uint64_t n = 2 * l;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_words; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t w = 0;
        for (int b = 0; b < 32; ++b) { const uint64_t i = k * 32 + b; w = w << 2 | (i < n ? text_at(fwd, l, i) : 0u); }
        W[k] = w;
    }
}

// the 32 bases from text position p on (zeros -- 'A' -- beyond the end) as one key
__device__ inline uint64_t key32(const uint64_t* W, uint64_t p)
{
    const uint64_t idx = p >> 5; const int s = (int)(p & 31) << 1;
    const uint64_t a = W[idx];
    return s ? a << s | W[idx + 1] >> (64 - s) : a;
}

// positions whose first two bases are bucket b (any order: the refinement below leaves no ties, so the suffix array does not
// depend on it): one atomic per wavefront
__global__ void k_bucket_collect(const uint64_t* W, uint64_t n, int b, uint64_t* pos, unsigned long long* count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {       // (a grid of n threads would exceed the 2^32 work-items of a launch)
        const uint64_t i = i0 + threadIdx.x;
        bool hit = false;
        if (i < n) {
            const uint64_t w = W[i >> 5]; const int s = (int)(i & 31) << 1;
            const uint32_t c0 = (uint32_t)(w >> (62 - s)) & 3u;
            const uint32_t c1 = i + 1 < n ? (uint32_t)(key32(W, i + 1) >> 62) : 0u;     // (beyond the end counts as 'A', as in the keys)
            hit = (int)(c0 * 4 + c1) == b;
        }
        const unsigned long long m = __ballot(hit);
        if (!m) continue;
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, (unsigned long long)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (hit) pos[base + __popcll(m & ((1ull << lane) - 1))] = i;
    }
}

__global__ void k_first_keys(const uint64_t* W, const uint64_t* pos, uint32_t m, uint64_t* key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) key[i] = key32(W, pos[i] + 2);
}

// tied[i] = element i shares its key with a neighbour; start[i] = i if it opens a run, else 0 (for the running maximum that
// turns into the run's id)
__global__ void k_mark_runs(const uint64_t* key, uint32_t m, uint8_t* tied, uint32_t* start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool same_prev = i > 0 && key[i] == key[i - 1], same_next = i + 1 < m && key[i] == key[i + 1];
    tied[i] = same_prev || same_next;
    start[i] = same_prev ? 0u : i;
}

__global__ void k_iota(uint32_t m, uint32_t* out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < m) out[i] = i; }

// the tied elements of a bucket: their positions and run ids, by slot
__global__ void k_gather_tied(const uint32_t* slots, uint32_t t, const uint64_t* order, const uint32_t* run, uint64_t* p, uint32_t* grp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t) { p[i] = order[slots[i]]; grp[i] = run[slots[i]]; }
}

// next 32 bases of the tied suffixes; `sec` orders those that have run out of text (the one starting later first)
__global__ void k_refine_keys(const uint64_t* W, uint64_t n, const uint64_t* p, uint32_t t, uint64_t off, uint64_t* key, uint64_t* sec, int* any_beyond)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t) return;
    const uint64_t q = p[i] + off;
    const bool beyond = q >= n;
    key[i] = beyond ? 0ull : key32(W, q);
    sec[i] = beyond ? n - p[i] : 1ull << 40;
    if (beyond) *any_beyond = 1;
}

template <typename T>
__global__ void k_permute(const uint32_t* perm, uint32_t t, const T* in, T* out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < t) out[i] = in[perm[i]]; }

// after the round's sort by (group, key, sec): write the positions back into their slots, and mark what is still tied
__global__ void k_after_round(const uint32_t* slots, uint32_t t, const uint64_t* p, const uint32_t* grp, const uint64_t* key, const uint64_t* sec,
                              uint64_t* order, uint8_t* tied, uint32_t* start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t) return;
    order[slots[i]] = p[i];
    const uint64_t far = 1ull << 40;
    auto eq = [&](uint32_t a, uint32_t b) { return grp[a] == grp[b] && key[a] == key[b] && !(sec[a] != far && sec[b] != far); };
    const bool same_prev = i > 0 && eq(i, i - 1), same_next = i + 1 < t && eq(i, i + 1);
    tied[i] = same_prev || same_next;
    start[i] = same_prev ? 0u : slots[i];
}
__global__ void k_compact_tied(const uint32_t* keep, uint32_t t2, const uint32_t* slots, const uint64_t* p, const uint32_t* newg, uint32_t* slots2, uint64_t* p2, uint32_t* grp2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t2) { const uint32_t k = keep[i]; slots2[i] = slots[k]; p2[i] = p[k]; grp2[i] = newg[k]; }
}

// ---- suffix array -> BWT / occ / sampled SA
__global__ void k_find_primary(const uint64_t* sa, uint64_t n, unsigned long long* primary)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (sa[i] == 0) *primary = i + 1;                    // rank k = i + 1 (rank 0 is the empty suffix)
}
// BWT symbols without the sentinel's: B[w], w = k - (k > primary), for the sentinel-inclusive ranks k != primary
__global__ void k_bwt_symbols(const uint8_t* fwd, uint64_t l, const uint64_t* sa, uint64_t n, uint64_t primary, uint8_t* B)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (uint64_t)gridDim.x * blockDim.x) {
        if (k == primary) continue;
        const uint64_t pos = k == 0 ? n : sa[k - 1];
        B[k - (k > primary)] = (uint8_t)text_at(fwd, l, pos - 1);
    }
}
// per 128-symbol block: the number of A, C, G, T in it (for the exclusive scans) and its eight packed words
__global__ void k_block_counts(const uint8_t* B, uint64_t n, uint64_t n_blk, uint64_t* cnt /* [4][n_blk] */, uint32_t* words /* [n_blk][8] */)
{
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blk; blk += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t c[4] = {0, 0, 0, 0};
    for (int wv = 0; wv < 8; ++wv) {
        uint32_t x = 0;
        for (int b = 0; b < 16; ++b) {
            const uint64_t i = blk * 128 + wv * 16 + b;
            if (i < n) { const uint32_t s = B[i]; x |= s << ((15 - b) << 1); ++c[s]; }
        }
        words[blk * 8 + wv] = x;
    }
    for (int s = 0; s < 4; ++s) cnt[(uint64_t)s * n_blk + blk] = c[s];
    }
}
// the .bwt body: per block 4 x u64 counts before the block, then its symbol words (only those that hold symbols); the totals at the end
__global__ void k_interleave(const uint64_t* excl /* [4][n_blk] */, const uint32_t* words, uint64_t n, uint64_t n_blk, uint32_t* out)
{
    for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blk; blk += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t* o = out + blk * 16;
    for (int s = 0; s < 4; ++s) { const uint64_t v = excl[(uint64_t)s * n_blk + blk]; o[2 * s] = (uint32_t)v; o[2 * s + 1] = (uint32_t)(v >> 32); }
    const uint64_t n_sym_words = (n + 15) / 16;
    for (int wv = 0; wv < 8; ++wv) if (blk * 8 + wv < n_sym_words) o[8 + wv] = words[blk * 8 + wv];
    }
}
__global__ void k_sample_sa(const uint64_t* sa, uint64_t n_sa, uint64_t* out)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_sa; j += (uint64_t)gridDim.x * blockDim.x)
        out[j] = j == 0 ? (uint64_t)-1 : sa[j * 32 - 1];
}

// (the kernels over the whole text loop with the grid's stride: a launch holds fewer than 2^32 work-items)
inline dim3 grid_for(uint64_t n, int block) { const uint64_t g = (n + block - 1) / block, cap = (uint64_t)1 << 23; return dim3((unsigned)(g < cap ? g : cap)); }   // 2^23 blocks of 256 cover any bucket (< 2^31 items) in one go

struct MaxOp { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

}  // namespace

bool device_index_available()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// fwd: the forward strand, one code 0..3 per base (ambiguous bases already replaced).  Fills primary, L2, seq_len, bwt, sa, sa_intv.
bool device_index_pieces(const std::vector<uint8_t>& fwd_h, IndexPieces& out, std::string* err)
{
    const uint64_t l = fwd_h.size(), n = 2 * l;
    if (l == 0) { if (err) *err = "empty reference"; return false; }
    Buf fwd, Wb, sa;
    if (!fwd.ensure(l + 64) || !Wb.ensure(((n + 31) / 32 + 4) * 8) || !sa.ensure(n * 8 + 64)) { if (err) *err = "out of device memory"; return false; }
    IDX_OK(hipMemcpy(fwd.p, fwd_h.data(), l, hipMemcpyHostToDevice));
    const uint64_t n_words = (n + 31) / 32 + 4;
    hipLaunchKernelGGL(k_pack_words, grid_for(n_words, 256), dim3(256), 0, 0, fwd.as<uint8_t>(), l, n_words, Wb.as<uint64_t>());
    const uint64_t* W = Wb.as<uint64_t>();
    uint64_t* SA = sa.as<uint64_t>();

    Buf cnt, tmp, key[2], val[2], tied, start, run, idx, slots[2], tp[2], tg[2], tkey[2], tsec[2], perm[2], nsel, flag;
    if (!cnt.ensure(64) || !nsel.ensure(64) || !flag.ensure(64)) { if (err) *err = "out of device memory"; return false; }
    auto need_tmp = [&](size_t b) { return tmp.ensure(b); };
    uint64_t at = 0;
    for (int b = 0; b < 16; ++b) {
        // ---- the bucket's positions, straight into their stretch of the suffix array
        IDX_OK(hipMemset(cnt.p, 0, 8));
        // (the count is not known beforehand: collect into a scratch of the largest possible size only once it is known)
        // first pass counts, second pass writes: two cheap passes over the packed text
        unsigned long long m64 = 0;
        {
            // counting pass: same kernel with a null destination would need a branch; use a small trick -- collect into SA + at
            // directly (the stretch [at, at + m) belongs to this bucket, and nothing beyond `at` has been written yet)
            hipLaunchKernelGGL(k_bucket_collect, grid_for(n, 256), dim3(256), 0, 0, W, n, b, SA + at, cnt.as<unsigned long long>());
            IDX_OK(hipGetLastError());
            IDX_OK(hipMemcpy(&m64, cnt.p, 8, hipMemcpyDeviceToHost));
        }
        if (m64 == 0) continue;
        if (m64 >= (1ull << 31)) { if (err) *err = "a two-base bucket holds 2^31 suffixes or more: reference too large for this builder"; return false; }
        const uint32_t m = (uint32_t)m64;
        uint64_t* order = SA + at;
        if (!key[0].ensure((size_t)m * 8) || !key[1].ensure((size_t)m * 8) || !val[1].ensure((size_t)m * 8)) { if (err) *err = "out of device memory"; return false; }
        hipLaunchKernelGGL(k_first_keys, grid_for(m, 256), dim3(256), 0, 0, W, order, m, key[0].as<uint64_t>());
        size_t tb = 0;
        IDX_OK(rocprim::radix_sort_pairs(nullptr, tb, key[0].as<uint64_t>(), key[1].as<uint64_t>(), order, val[1].as<uint64_t>(), m, 0, 64, (hipStream_t)0));
        if (!need_tmp(tb)) { if (err) *err = "out of device memory"; return false; }
        IDX_OK(rocprim::radix_sort_pairs(tmp.p, tb, key[0].as<uint64_t>(), key[1].as<uint64_t>(), order, val[1].as<uint64_t>(), m, 0, 64, (hipStream_t)0));
        IDX_OK(hipMemcpyAsync(order, val[1].p, (size_t)m * 8, hipMemcpyDeviceToDevice, 0));
        // ---- what is still tied after the first key
        if (!tied.ensure(m) || !start.ensure((size_t)m * 4) || !run.ensure((size_t)m * 4) || !idx.ensure((size_t)m * 4) || !slots[0].ensure((size_t)m * 4)) { if (err) *err = "out of device memory"; return false; }
        hipLaunchKernelGGL(k_mark_runs, grid_for(m, 256), dim3(256), 0, 0, key[1].as<uint64_t>(), m, tied.as<uint8_t>(), start.as<uint32_t>());
        IDX_OK(rocprim::inclusive_scan(nullptr, tb, start.as<uint32_t>(), run.as<uint32_t>(), m, MaxOp(), (hipStream_t)0));
        if (!need_tmp(tb)) { if (err) *err = "out of device memory"; return false; }
        IDX_OK(rocprim::inclusive_scan(tmp.p, tb, start.as<uint32_t>(), run.as<uint32_t>(), m, MaxOp(), (hipStream_t)0));
        hipLaunchKernelGGL(k_iota, grid_for(m, 256), dim3(256), 0, 0, m, idx.as<uint32_t>());
        IDX_OK(rocprim::select(nullptr, tb, idx.as<uint32_t>(), tied.as<uint8_t>(), slots[0].as<uint32_t>(), nsel.as<uint32_t>(), m, (hipStream_t)0));
        if (!need_tmp(tb)) { if (err) *err = "out of device memory"; return false; }
        IDX_OK(rocprim::select(tmp.p, tb, idx.as<uint32_t>(), tied.as<uint8_t>(), slots[0].as<uint32_t>(), nsel.as<uint32_t>(), m, (hipStream_t)0));
        uint32_t t = 0;
        IDX_OK(hipMemcpy(&t, nsel.p, 4, hipMemcpyDeviceToHost));
        int cur = 0;
        if (t) {
            for (int k = 0; k < 2; ++k)
                if (!tp[k].ensure((size_t)t * 8) || !tg[k].ensure((size_t)t * 4) || !tkey[k].ensure((size_t)t * 8) || !tsec[k].ensure((size_t)t * 8) || !perm[k].ensure((size_t)t * 4) || !slots[1].ensure((size_t)t * 4))
                    { if (err) *err = "out of device memory"; return false; }
            hipLaunchKernelGGL(k_gather_tied, grid_for(t, 256), dim3(256), 0, 0, slots[0].as<uint32_t>(), t, order, run.as<uint32_t>(), tp[0].as<uint64_t>(), tg[0].as<uint32_t>());
        }
        uint64_t off = 34;
        while (t) {
            // keys of this round; sort by (group, key, sec): stable passes, least significant first
            IDX_OK(hipMemset(flag.p, 0, 4));
            hipLaunchKernelGGL(k_refine_keys, grid_for(t, 256), dim3(256), 0, 0, W, n, tp[cur].as<uint64_t>(), t, off, tkey[0].as<uint64_t>(), tsec[0].as<uint64_t>(), flag.as<int>());
            int any_beyond = 0;
            IDX_OK(hipMemcpy(&any_beyond, flag.p, 4, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(k_iota, grid_for(t, 256), dim3(256), 0, 0, t, perm[0].as<uint32_t>());
            int pc = 0;
            auto sort_by = [&](auto* keys_in, auto* keys_scratch, int bits) -> bool {     // perm[pc] <- stable order of keys_in[perm[pc]]
                using K = std::remove_pointer_t<decltype(keys_in)>;
                hipLaunchKernelGGL((k_permute<K>), grid_for(t, 256), dim3(256), 0, 0, perm[pc].as<uint32_t>(), t, (const K*)keys_in, keys_scratch);
                size_t bytes = 0;
                Buf& sorted_keys = sizeof(K) == 8 ? key[0] : idx;      // (free at this point: the bucket's first-key buffers)
                if (rocprim::radix_sort_pairs(nullptr, bytes, keys_scratch, sorted_keys.as<K>(), perm[pc].as<uint32_t>(), perm[pc ^ 1].as<uint32_t>(), t, 0, bits, (hipStream_t)0) != hipSuccess) return false;
                if (!need_tmp(bytes)) return false;
                if (rocprim::radix_sort_pairs(tmp.p, bytes, keys_scratch, sorted_keys.as<K>(), perm[pc].as<uint32_t>(), perm[pc ^ 1].as<uint32_t>(), t, 0, bits, (hipStream_t)0) != hipSuccess) return false;
                pc ^= 1;
                return true;
            };
            if (any_beyond && !sort_by(tsec[0].as<uint64_t>(), tsec[1].as<uint64_t>(), 41)) { if (err) *err = "device sort failed"; return false; }
            if (!sort_by(tkey[0].as<uint64_t>(), tkey[1].as<uint64_t>(), 64)) { if (err) *err = "device sort failed"; return false; }
            if (!sort_by(tg[cur].as<uint32_t>(), tg[cur ^ 1].as<uint32_t>(), 32)) { if (err) *err = "device sort failed"; return false; }
            // apply the permutation
            hipLaunchKernelGGL((k_permute<uint64_t>), grid_for(t, 256), dim3(256), 0, 0, perm[pc].as<uint32_t>(), t, (const uint64_t*)tp[cur].as<uint64_t>(), tp[cur ^ 1].as<uint64_t>());
            hipLaunchKernelGGL((k_permute<uint64_t>), grid_for(t, 256), dim3(256), 0, 0, perm[pc].as<uint32_t>(), t, (const uint64_t*)tkey[0].as<uint64_t>(), tkey[1].as<uint64_t>());
            hipLaunchKernelGGL((k_permute<uint64_t>), grid_for(t, 256), dim3(256), 0, 0, perm[pc].as<uint32_t>(), t, (const uint64_t*)tsec[0].as<uint64_t>(), tsec[1].as<uint64_t>());
            hipLaunchKernelGGL((k_permute<uint32_t>), grid_for(t, 256), dim3(256), 0, 0, perm[pc].as<uint32_t>(), t, (const uint32_t*)tg[cur].as<uint32_t>(), tg[cur ^ 1].as<uint32_t>());
            cur ^= 1;
            // slots stay ascending and a group stays inside its slot range: positions back into place, then the new ties
            hipLaunchKernelGGL(k_after_round, grid_for(t, 256), dim3(256), 0, 0, slots[0].as<uint32_t>(), t, tp[cur].as<uint64_t>(), tg[cur].as<uint32_t>(), tkey[1].as<uint64_t>(), tsec[1].as<uint64_t>(),
                               order, tied.as<uint8_t>(), start.as<uint32_t>());
            size_t bytes = 0;
            IDX_OK(rocprim::inclusive_scan(nullptr, bytes, start.as<uint32_t>(), run.as<uint32_t>(), t, MaxOp(), (hipStream_t)0));
            if (!need_tmp(bytes)) { if (err) *err = "out of device memory"; return false; }
            IDX_OK(rocprim::inclusive_scan(tmp.p, bytes, start.as<uint32_t>(), run.as<uint32_t>(), t, MaxOp(), (hipStream_t)0));
            hipLaunchKernelGGL(k_iota, grid_for(t, 256), dim3(256), 0, 0, t, idx.as<uint32_t>());
            IDX_OK(rocprim::select(nullptr, bytes, idx.as<uint32_t>(), tied.as<uint8_t>(), perm[0].as<uint32_t>(), nsel.as<uint32_t>(), t, (hipStream_t)0));
            if (!need_tmp(bytes)) { if (err) *err = "out of device memory"; return false; }
            IDX_OK(rocprim::select(tmp.p, bytes, idx.as<uint32_t>(), tied.as<uint8_t>(), perm[0].as<uint32_t>(), nsel.as<uint32_t>(), t, (hipStream_t)0));
            uint32_t t2 = 0;
            IDX_OK(hipMemcpy(&t2, nsel.p, 4, hipMemcpyDeviceToHost));
            if (t2) {
                hipLaunchKernelGGL(k_compact_tied, grid_for(t2, 256), dim3(256), 0, 0, perm[0].as<uint32_t>(), t2, slots[0].as<uint32_t>(), tp[cur].as<uint64_t>(), run.as<uint32_t>(),
                                   slots[1].as<uint32_t>(), tp[cur ^ 1].as<uint64_t>(), tg[cur ^ 1].as<uint32_t>());
                IDX_OK(hipMemcpyAsync(slots[0].p, slots[1].p, (size_t)t2 * 4, hipMemcpyDeviceToDevice, 0));
                cur ^= 1;
            }
            t = t2;
            off += 32;
            if (off > n + 64) { if (err) *err = "internal error: suffix ties did not resolve"; return false; }
        }
        at += m;
    }
    IDX_OK(hipDeviceSynchronize());
    if (at != n) { if (err) *err = "internal error: buckets do not cover the text"; return false; }
    for (Buf* f : { &key[0], &key[1], &val[1], &tied, &start, &run, &idx, &slots[0], &slots[1], &tmp }) f->release();

    // ---- BWT, occ checkpoints, sampled SA
    out.seq_len = n;
    {   // L2 from the forward strand's base counts (the reverse complement mirrors them)
        uint64_t c[4] = {0, 0, 0, 0};
        for (uint8_t x : fwd_h) ++c[x];
        const uint64_t t4[4] = { c[0] + c[3], c[1] + c[2], c[2] + c[1], c[3] + c[0] };
        out.L2[0] = 0;
        for (int s = 0; s < 4; ++s) out.L2[s + 1] = out.L2[s] + t4[s];
    }
    IDX_OK(hipMemset(cnt.p, 0, 8));
    hipLaunchKernelGGL(k_find_primary, grid_for(n, 256), dim3(256), 0, 0, SA, n, cnt.as<unsigned long long>());
    unsigned long long primary = 0;
    IDX_OK(hipMemcpy(&primary, cnt.p, 8, hipMemcpyDeviceToHost));
    out.primary = primary;
    Buf B, bc, be, words, bwt, sas;
    const uint64_t n_blk = (n + 127) / 128, n_occ = n_blk + 1;
    const uint64_t n_bwt = (n + 15) / 16 + n_occ * 8;
    if (!B.ensure(n + 64) || !bc.ensure(4 * n_blk * 8) || !be.ensure(4 * n_blk * 8) || !words.ensure(n_blk * 8 * 4) || !bwt.ensure((n_blk * 16 + 16) * 4)) { if (err) *err = "out of device memory"; return false; }
    hipLaunchKernelGGL(k_bwt_symbols, grid_for(n + 1, 256), dim3(256), 0, 0, fwd.as<uint8_t>(), l, SA, n, (uint64_t)primary, B.as<uint8_t>());
    hipLaunchKernelGGL(k_block_counts, grid_for(n_blk, 128), dim3(128), 0, 0, B.as<uint8_t>(), n, n_blk, bc.as<uint64_t>(), words.as<uint32_t>());
    uint64_t totals[4];
    for (int s = 0; s < 4; ++s) {
        size_t bytes = 0;
        IDX_OK(rocprim::exclusive_scan(nullptr, bytes, bc.as<uint64_t>() + (uint64_t)s * n_blk, be.as<uint64_t>() + (uint64_t)s * n_blk, (uint64_t)0, n_blk, rocprim::plus<uint64_t>(), (hipStream_t)0));
        if (!tmp.ensure(bytes)) { if (err) *err = "out of device memory"; return false; }
        IDX_OK(rocprim::exclusive_scan(tmp.p, bytes, bc.as<uint64_t>() + (uint64_t)s * n_blk, be.as<uint64_t>() + (uint64_t)s * n_blk, (uint64_t)0, n_blk, rocprim::plus<uint64_t>(), (hipStream_t)0));
        uint64_t last_e = 0, last_c = 0;
        IDX_OK(hipMemcpy(&last_e, be.as<uint64_t>() + (uint64_t)s * n_blk + n_blk - 1, 8, hipMemcpyDeviceToHost));
        IDX_OK(hipMemcpy(&last_c, bc.as<uint64_t>() + (uint64_t)s * n_blk + n_blk - 1, 8, hipMemcpyDeviceToHost));
        totals[s] = last_e + last_c;
    }
    IDX_OK(hipMemset(bwt.p, 0, (n_blk * 16 + 16) * 4));
    hipLaunchKernelGGL(k_interleave, grid_for(n_blk, 128), dim3(128), 0, 0, be.as<uint64_t>(), words.as<uint32_t>(), n, n_blk, bwt.as<uint32_t>());
    out.bwt.assign(n_bwt, 0);
    IDX_OK(hipMemcpy(out.bwt.data(), bwt.p, (n_bwt - 8) * 4, hipMemcpyDeviceToHost));
    memcpy(&out.bwt[n_bwt - 8], totals, 32);
    out.sa_intv = 32;
    const uint64_t n_sa = (n + 32) / 32;
    if (!sas.ensure(n_sa * 8)) { if (err) *err = "out of device memory"; return false; }
    hipLaunchKernelGGL(k_sample_sa, grid_for(n_sa, 256), dim3(256), 0, 0, SA, n_sa, sas.as<uint64_t>());
    out.sa.assign(n_sa, 0);
    IDX_OK(hipMemcpy(out.sa.data(), sas.p, n_sa * 8, hipMemcpyDeviceToHost));
    IDX_OK(hipGetLastError());
    return true;
}
