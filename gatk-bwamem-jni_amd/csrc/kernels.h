// kernels.h -- host-callable launchers of the HIP kernels (one translation unit per stage).
#pragma once
#include <hip/hip_runtime.h>
#include "bwamem_types.h"

void launch_build_occ64(hipStream_t st, const uint32_t* bwt, uint64_t n_blocks, uint4* occ);
// suffix array at every ix.sa_intv-th rank (lo/hi, (seq_len >> sa_shift) + 1 entries) from the image's sampling; *err: device int, OR-ed on failure
void launch_sa_densify(hipStream_t st, const DevIndex& ix, const uint64_t* sa_src, uint64_t n_src, int src_intv, uint32_t* lo, uint8_t* hi, int32_t* err);
void launch_encode(hipStream_t st, uint8_t* seq, int64_t n_bytes);
void launch_unpack_pac(hipStream_t st, const DevIndex& ix, int64_t start, int64_t n, uint8_t* dst);
void launch_seed(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
size_t scan_tmp_bytes(int64_t n);
void launch_scan(hipStream_t st, const int32_t* in, int64_t* out, int n, int64_t* tmp);   // tmp: scan_tmp_bytes(n), or null (one-workgroup form)
void launch_order(hipStream_t st, const int32_t* n_seeds, int n, int32_t* bins64, int32_t* order);   // TileView::order
size_t nul_tmp_bytes(int64_t n_bytes);
void launch_nul_offsets(hipStream_t st, const uint8_t* seq, int64_t n_bytes, int64_t* off, int64_t n_reads_max, int64_t* n_found, void* tmp);
void launch_sa(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int64_t n_occ);
void launch_chain(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, Chain* chain_store);
// seed re-scoring (long reads): jobs/results sized by pe_rescue_bytes for `cap` >= the tile's seed count; first_num = 2 x n_reads ints, cnt = 1 int
bool rescore_needed(const MemOpt& opt, const TileView& tv);
void launch_rescore(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, void* jobs, void* results, int32_t* first_num, int32_t* cnt, int cap);
size_t extend_lds_bytes(const MemOpt& opt, int max_len);       // dynamic LDS k_extend asks for; the caller checks it against the device limit
void launch_extend(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_post1(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_final_prep(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
size_t gcigar_slab_bytes(const MemOpt& opt, int max_len);      // traceback slab per resident workgroup of the wave form (long reads), 0 = none
int gcigar_slab_grid(const DevIndex& ix, int n_jobs);
void launch_gcigar(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int n_jobs, const void* jobs, void* outs, uint32_t* cig_pool, int cig_cap,
                   uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur, uint8_t* slabs, size_t slab_bytes, int* queue);
void launch_final_se(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, const void* job_out, const uint32_t* job_cig, int cig_cap);
void launch_pack(hipStream_t st, const TileView& tv, uint8_t* dst);

// paired-end path (k_pe.hip)
void launch_pestat_cand(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int8_t* cand_dir, int64_t* cand_is);
void launch_pe_caps(hipStream_t st, const MemOpt& opt, const TileView& tv, int32_t* caps);
void launch_pe_copy_regs(hipStream_t st, const TileView& tv, const AlnReg* src, const int64_t* src_off, AlnReg* dst, const int64_t* dst_off, const int32_t* n_regs);
void launch_pe_pair(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, AlnReg* regs, const int64_t* reg_off,
                    int32_t* n_regs, int32_t* ints, void* vpool, void* keys, uint8_t* scratch, int64_t scratch_per_pair, int cap_h, int cap_b, int cap_u, const MemPestat* pes, const PairTab& ptab, void* states,
                    void* rescue_jobs, void* rescue_res, int32_t* rescue_first, int32_t* rescue_num, int32_t* rescue_cnt, int rescue_cap);
size_t pe_rescue_bytes(int what, int cap);       // what = 0: SwJob[cap], 1: KswR[cap]
void launch_sw_jobs(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, const void* jobs, const int32_t* cnt, int cap, void* results, int cap_b, int max_qlen);
void launch_pe_out(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, AlnReg* regs, const int64_t* reg_off,
                   int32_t* n_regs, int32_t* ints, const MemPestat* pes, const void* states, const void* job_out, const uint32_t* job_cig, int cig_cap);
size_t pe_state_bytes(int n_reads);
