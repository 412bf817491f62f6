/*
 * bwamem_hip.h -- C ABI of the MI355X-native BWA-MEM library (libbwamem_hip.so).
 *
 * The first block is the drop-in boundary: the same JNI-free entry points the reference's
 * JNI glue binds (reference: src/main/c/jnibwa.h:11-16, defined in src/main/c/jnibwa.c), with
 * identical argument meaning, ownership and error behaviour.  A maintainer of
 * broadinstitute/gatk-bwamem-jni links org_broadinstitute_hellbender_utils_bwa_BwaMemIndex.c
 * and init.c against this library instead of jnibwa.o + libbwa.a.
 * Handles are opaque; plain pointers and sizes only.
 */
#ifndef BWAMEM_HIP_H_
#define BWAMEM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bwaidx_s bwaidx_t;      /* opaque; the jlong handle of BwaMemIndex.java:83 */
typedef struct mem_opt_s mem_opt_t;    /* 168 bytes, offsets pinned by BwaMemAligner.java:46-138 */
typedef struct mem_pestat_s mem_pestat_t; /* {int low, high, failed; double avg, std;}, 32 bytes x 4 orientations */

/* replaces bwa_idx_build() as called from ...BwaMemIndex.c:59 (declared, never defined, at jnibwa.h:11).
 * algo: "auto", "is", "rb2" select nothing here (one builder); any other name returns -1.  0 = ok. */
int jnibwa_createReferenceIndex(const char* refFileName, const char* indexPrefix, const char* algoName);

/* jnibwa.c:126-152: <prefix>.{amb,ann,bwt,pac,sa}[,alt] -> one contiguous image file.  0 ok, 2 I/O error. */
int jnibwa_createIndexFile(const char* refName, const char* imgName);

/* jnibwa.c:154-165: takes ownership of fd; mmaps the image read-only, uploads bwt/occ, SA and pac
 * to HBM.  Returns 0 on failure (Java then throws CouldNotReadImageException, BwaMemIndex.java:334). */
bwaidx_t* jnibwa_openIndex(int fd);

/* jnibwa.c:167-172 */
int jnibwa_destroyIndex(bwaidx_t* pIdx);

/* jnibwa.c:174-195: int32 n, then (int32 len, bytes) per contig; free with jnibwa_free */
void* jnibwa_getRefContigNames(bwaidx_t* pIdx, size_t* pBufSize);

/* jnibwa.c:197-235: pSeq = uint32 nSeqs + nSeqs NUL-terminated base strings (left untouched here; upstream
 * overwrites the bases with 0..4 codes, which Java never observes: BwaMemAligner.java:203-210); peStats = mem_pestat_t[4] or NULL (infer).  Returns a malloc'ed
 * int32 stream in the layout of ...BwaMemIndex.c:115-141, or NULL on any device error. */
void* jnibwa_createAlignments(bwaidx_t* pIdx, mem_opt_t* pOpts, mem_pestat_t* peStats, char* pSeq, size_t* pBufSize);

/* mem_opt_init() as wrapped at ...BwaMemIndex.c:89-92; free with jnibwa_free */
mem_opt_t* jnibwa_createDefaultOptions(void);

/* the one allocator behind destroyByteBuffer (...BwaMemIndex.c:157-160) */
void jnibwa_free(void* p);

/* ...BwaMemIndex.c:163-165 */
const char* jnibwa_getVersion(void);

/* ------------------------------------------------------------------------------------------------
 * Additive device-level entry points (bench, multi-GPU drivers).  Not part of the reference ABI. */

int bwamem_hip_set_device(int device);        /* indexes opened afterwards live on this one device (one process per GPU: bench.py's ranks) */
int bwamem_hip_device_count(void);
/* jnibwa_openIndex puts a replica of the index on every device named by BWAMEM_HIP_DEVICES ("all", or e.g. "0,1,2,3"; unset: the
 * device of bwamem_hip_set_device if that was called, else all visible devices); jnibwa_createAlignments (jnibwa.c:197-235: one
 * native call per batch) then cuts a large call across the replicas and sends small concurrent calls to them in turn.
 * -> the number of replicas behind the handle. */
int bwamem_hip_index_replicas(bwaidx_t* idx);

/* Tooling (bench.py --image): the contig lengths of an open index (returns the number of contigs; lens may be NULL), and
 * n bases of its packed reference from position start, one code 0..3 per byte, into DEVICE memory d_dst.  0 = ok. */
int bwamem_hip_index_contig_lengths(bwaidx_t* idx, int64_t* lens, int cap);
int bwamem_hip_index_unpack_pac(bwaidx_t* idx, int64_t start, int64_t n, void* d_dst);

/* Tooling (bench.py): an index image straight from base codes (0..3, one per byte, host memory) and a contig table, built on
 * the device (the same builder jnibwa_createReferenceIndex uses when a device is visible).  0 = ok. */
int bwamem_hip_build_image(const uint8_t* codes, int64_t l_pac, int32_t n_contigs, const char* const* names, const int64_t* lens, const char* img_path);

typedef struct bwamem_batch_s bwamem_batch_t; /* a request resident in HBM */

/* upload a request buffer (same wire format as pSeq above); the host buffer is left untouched */
bwamem_batch_t* bwamem_hip_batch_upload(bwaidx_t* idx, const char* pSeq, size_t nBytes);
/* the same for a payload that already lives in HBM: d_payload = the NUL-terminated base strings (without the
 * leading count), h_offsets = nReads+1 host offsets of the reads within it (h_offsets[nReads] = nBytes) */
bwamem_batch_t* bwamem_hip_batch_wrap_device(bwaidx_t* idx, const void* d_payload, size_t nBytes, uint32_t nReads, const int64_t* h_offsets);
/* run the whole hot path; results stay in HBM.  read_id0 = index of the first read within the
 * logical call (shards of one call must carry their global base index).  0 = ok. */
int bwamem_hip_batch_align(bwaidx_t* idx, const mem_opt_t* opt, const mem_pestat_t* pes, bwamem_batch_t* b, int64_t read_id0);
/* A paired-end call in two steps, for callers that shard ONE logical createAlignments call over several devices or
 * ranks (caveat 2).  With inferred insert-size statistics (pes == NULL at jnibwa.c:214) upstream's
 * mem_pestat is a reduction over all pairs of the call, between region finding and pairing; a shard must therefore
 * stop after phase 1, hand out its per-pair (orientation, insert size) candidates, and finish with the statistics
 * of the whole call:
 *   _pe_begin       phase 1 (seeding .. regions) of this shard's reads; opt must carry MEM_F_PE (0x2 at offset 60)
 *   _pe_candidates  n = pairs of the shard; dir[i] = orientation 0..3 or -1 (no candidate), isize[i] = insert size;
 *                   dir == NULL: just returns n
 *   bwamem_hip_pestat  upstream mem_pestat's reduction over candidates gathered from all shards (order-independent)
 *   _pe_finish      phase 2 (mate rescue, pairing, records) with those statistics; then download as usual
 * bwamem_hip_batch_align does the same in one call when the batch is the whole logical call. */
int bwamem_hip_batch_pe_begin(bwaidx_t* idx, const mem_opt_t* opt, bwamem_batch_t* b, int64_t read_id0);
size_t bwamem_hip_batch_pe_candidates(const bwamem_batch_t* b, int8_t* dir, int64_t* isize);
void bwamem_hip_pestat(const mem_opt_t* opt, const int8_t* dir, const int64_t* isize, size_t n, mem_pestat_t* pes /* [4] */);
int bwamem_hip_batch_pe_finish(bwaidx_t* idx, const mem_opt_t* opt, const mem_pestat_t* pes /* [4] */, bwamem_batch_t* b);
size_t bwamem_hip_batch_result_bytes(const bwamem_batch_t* b);
int bwamem_hip_batch_download(bwamem_batch_t* b, void* dst);
void bwamem_hip_batch_free(bwamem_batch_t* b);

/* SAM text on the native side (additive).  pSeq = the request handed to jnibwa_createAlignments,
 * response = what it returned.  One line per record, layout described in csrc/sam_writer.cpp; readNames = nSeqs names or NULL
 * ("r<index>", paired: "p<pair index>"); paired = the call carried MEM_F_PE.  Both return malloc'ed, NUL-terminated text (free
 * with jnibwa_free) or NULL when the response does not parse against the request. */
char* bwamem_hip_sam_header(bwaidx_t* idx, size_t* pBytes);
char* bwamem_hip_response_to_sam(bwaidx_t* idx, const char* pSeq, const void* response, size_t responseBytes, const char* const* readNames, int paired, size_t* pBytes);

typedef struct {
    /* algorithmic counters */
    uint64_t n_reads, n_ext, n_lf, n_sa, n_dp_cells;
    /* accumulated device time per kernel, ms, and launch counts (HIP events on the launch stream) */
    double ms_encode, ms_seed, ms_sa, ms_chain, ms_extend, ms_post, ms_final, ms_pack, ms_other;
    uint64_t n_launch_seed, n_launch_sa, n_launch_extend;
    uint64_t n_tiles, n_retries;
} bwamem_stats_t;

void bwamem_hip_stats_enable(int on);          /* per-kernel HIP-event timing (off by default) */
void bwamem_hip_stats_reset(void);
void bwamem_hip_stats_get(bwamem_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif
