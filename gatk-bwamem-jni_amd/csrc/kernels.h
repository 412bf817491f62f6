// kernels.h -- host-callable launchers of the HIP kernels (one translation unit per stage).
#pragma once
#include <hip/hip_runtime.h>
#include "bwamem_types.h"

void launch_encode(hipStream_t st, uint8_t* seq, int64_t n_bytes);
void launch_seed(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_scan(hipStream_t st, const int32_t* in, int64_t* out, int n);
void launch_sa(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int64_t n_occ);
void launch_chain(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, Chain* chain_store);
void launch_extend(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_post1(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_final_se(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv);
void launch_pack(hipStream_t st, const TileView& tv, uint8_t* dst);
