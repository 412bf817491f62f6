// index_io.h -- host-side index image: parse / assemble the contiguous ".img" layout that
// GATK users already have on disk (reference: jnibwa.c:126-165; format SURVEY.md App. A).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

struct ContigInfo {
    int64_t offset; int32_t len, n_ambs; uint32_t gi; int32_t is_alt;
    std::string name, anno;
};

struct HoleInfo { int64_t offset; int32_t len; char amb; };

struct HostIndex {
    const uint8_t* mem = nullptr; size_t l_mem = 0;
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0, bwt_size = 0, n_sa = 0;
    int sa_intv = 0;
    const uint32_t* bwt = nullptr;     // bwt_size words
    const uint64_t* sa = nullptr;      // n_sa values, sa[0] = -1
    int64_t l_pac = 0; uint32_t seed = 0;
    std::vector<ContigInfo> contigs;
    std::vector<HoleInfo> holes;
    const uint8_t* pac = nullptr;      // l_pac/4+1 bytes
};

// false if the image is malformed (every offset is derived from the embedded counts)
bool parse_index_image(const uint8_t* mem, size_t l_mem, HostIndex& out);

// the five index files (+ optional .alt) -> image bytes; empty vector on I/O or format error
std::vector<uint8_t> image_from_index_files(const std::string& prefix, std::string* err);

// in-memory pieces -> image bytes (used by the builder and by image_from_index_files)
struct IndexPieces {
    uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0}, seq_len = 0;
    std::vector<uint32_t> bwt;         // interleaved occ + bwt words
    int sa_intv = 32;
    std::vector<uint64_t> sa;          // n_sa entries incl. sa[0] = -1
    int64_t l_pac = 0; uint32_t seed = 11;
    std::vector<ContigInfo> contigs;
    std::vector<HoleInfo> holes;
    std::vector<uint8_t> pac;          // l_pac/4+1 bytes
};
std::vector<uint8_t> image_from_pieces(const IndexPieces& p);

// fasta -> <prefix>.{pac,ann,amb,bwt,sa}; replaces upstream bwa_idx_build (bwtindex.c).  The suffix array, BWT, occ checkpoints
// and SA samples are computed on the device when one is visible (k_index.hip), on the host otherwise or when
// BWAMEM_HIP_INDEX_BUILDER=host; the files are byte-identical either way.
bool build_index_files(const std::string& fasta, const std::string& prefix, std::string* err);
// k_index.hip: is there a device to build on / the device half of the builder (fwd: one code 0..3 per forward-strand base)
bool device_index_available();
bool device_index_pieces(const std::vector<uint8_t>& fwd, IndexPieces& out, std::string* err);
