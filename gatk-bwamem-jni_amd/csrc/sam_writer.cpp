// sam_writer.cpp -- SAM text on the native side (additive, not part of the reference ABI).
//
// GATK callers decode the response of jnibwa_createAlignments into BwaMemAlignment objects (BwaMemAligner.java:215-308) and
// re-encode those as SAM records themselves.  bwamem_hip_response_to_sam does that round trip natively: it walks the request
// (pSeq: count + NUL-terminated base strings, BwaMemAligner.java:198-209) and the response (layout of jnibwa.c:43-98) side by
// side and writes one SAM line per record.  The reference contains no SAM writer, so the line layout follows the SAM
// specification and, where the specification leaves a choice, upstream's mem_aln2sam (bwamem.c) as this repository restates it:
//   * FLAG is the record's flag (strand, mate, secondary and supplementary bits are already there: BwaMemIndexTest.java:84-127 pins
//     0x61/0x63/0x91/0x93 on the response itself);
//   * records after a read's first one turn soft clips into hard clips and carry only the aligned bases (upstream without -Y);
//   * SEQ is the read as sequenced for forward alignments, its reverse complement for reverse ones; QUAL is '*' (the request has none);
//   * RNEXT/PNEXT/TLEN from the record's mate fields ("=" for the same contig); unmapped reads of a pair take their mate's place;
//   * tags NM:i MD:Z AS:i XS:i and XA:Z when the record has them.
// Read names: supplied by the caller, else "r<index>" (paired: "p<pair index>" for both mates).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "index_io.h"
#include "../../include/bwamem_hip.h"

namespace {

struct Out {
    std::string s;
    void put(const char* p) { s += p; }
    void put(const std::string& p) { s += p; }
    void num(long long v) { char b[32]; snprintf(b, sizeof b, "%lld", v); s += b; }
    void ch(char c) { s.push_back(c); }
};

inline int32_t rd32(const uint8_t*& p) { int32_t v; memcpy(&v, p, 4); p += 4; return v; }

inline char comp(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}

}  // namespace

// the contig names and lengths of an open index (pipeline.cpp)
const std::vector<ContigInfo>& bwamem_index_contigs(const bwaidx_t* idx);

extern "C" {

char* bwamem_hip_sam_header(bwaidx_t* idx, size_t* pBytes)
{
    if (pBytes) *pBytes = 0;
    if (!idx) return 0;
    try {
        Out o;
        o.put("@HD\tVN:1.6\tSO:unsorted\tGO:query\n");
        for (const ContigInfo& c : bwamem_index_contigs(idx)) { o.put("@SQ\tSN:"); o.put(c.name); o.put("\tLN:"); o.num(c.len); o.ch('\n'); }
        o.put("@PG\tID:bwamem-hip\tPN:bwamem-hip\tVN:"); o.put(jnibwa_getVersion()); o.ch('\n');
        char* r = (char*)malloc(o.s.size() + 1);
        if (!r) return 0;
        memcpy(r, o.s.data(), o.s.size() + 1);
        if (pBytes) *pBytes = o.s.size();
        return r;
    } catch (...) { return 0; }
}

char* bwamem_hip_response_to_sam(bwaidx_t* idx, const char* pSeq, const void* response, size_t responseBytes, const char* const* readNames, int paired, size_t* pBytes)
{
    if (pBytes) *pBytes = 0;
    if (!idx || !pSeq || !response) return 0;
    try {
        const std::vector<ContigInfo>& contigs = bwamem_index_contigs(idx);
        uint32_t n_reads; memcpy(&n_reads, pSeq, 4);
        const char* q = pSeq + 4;
        const uint8_t* p = (const uint8_t*)response;
        const uint8_t* const end = p + responseBytes;
        Out o;
        o.s.reserve(responseBytes * 4 + (size_t)n_reads * 64);
        static const char OPS[] = "MIDNSHP=X";
        for (uint32_t r = 0; r < n_reads; ++r) {
            const size_t l_seq = strlen(q);
            const char* seq = q;
            q += l_seq + 1;
            std::string name;
            if (readNames && readNames[r]) name = readNames[r];
            else { char b[32]; snprintf(b, sizeof b, paired ? "p%u" : "r%u", paired ? r >> 1 : r); name = b; }
            if (paired && (n_reads & 1u) && r == n_reads - 1) continue;      // an odd last read of a paired call produces no bytes (jnibwa.c:214: n >> 1 pairs)
            if (p + 4 > end) return 0;
            const int32_t n_aln = rd32(p);
            for (int32_t k = 0; k < n_aln; ++k) {
                if (p + 4 > end) return 0;
                const int32_t fm = rd32(p);
                const int flag = (fm >> 16) & 0xffff, mapq = fm & 0xff;
                int32_t rid = -1, pos = -1, nm = 0, as = 0, xs = 0, n_cig = 0;
                std::vector<uint32_t> cig;
                std::string md, xa;
                if (!(flag & 4)) {
                    if (p + 24 > end) return 0;
                    rid = rd32(p); pos = rd32(p); nm = rd32(p); as = rd32(p); xs = rd32(p); n_cig = rd32(p);
                    if (n_cig < 0 || p + 4 * (size_t)n_cig + 4 > end) return 0;
                    cig.resize((size_t)n_cig);
                    for (int32_t c = 0; c < n_cig; ++c) cig[c] = (uint32_t)rd32(p);
                    const int32_t n_md = rd32(p);
                    if (n_md < 0 || p + ((n_md + 3) & ~3) + 4 > end) return 0;
                    md.assign((const char*)p, (size_t)n_md); p += (n_md + 3) & ~3;
                    const int32_t n_xa = rd32(p);
                    if (n_xa < 0 || p + ((n_xa + 3) & ~3) > end) return 0;
                    xa.assign((const char*)p, (size_t)n_xa); p += (n_xa + 3) & ~3;
                }
                int32_t mrid = -1, mpos = -1, tlen = 0;
                const bool has_mate = (flag & 9) == 1;
                if (has_mate) { if (p + 12 > end) return 0; mrid = rd32(p); mpos = rd32(p); tlen = rd32(p); }
                if (rid >= (int32_t)contigs.size() || mrid >= (int32_t)contigs.size()) return 0;
                // ---- the line
                const bool hard = k > 0 && !(flag & 4);             // later records of a read: clipped bases are not repeated
                o.put(name); o.ch('\t'); o.num(flag); o.ch('\t');
                if (rid >= 0) { o.put(contigs[rid].name); o.ch('\t'); o.num((long long)pos + 1); }
                else if (has_mate && mrid >= 0) { o.put(contigs[mrid].name); o.ch('\t'); o.num((long long)mpos + 1); }     // an unmapped mate sits at its mate's place
                else o.put("*\t0");
                o.ch('\t'); o.num(mapq); o.ch('\t');
                int clip5 = 0, clip3 = 0;
                if (cig.empty()) o.ch('*');
                else {
                    for (size_t c = 0; c < cig.size(); ++c) {
                        int op = (int)(cig[c] & 0xf); const uint32_t len = cig[c] >> 4;
                        if (op == 4 || op == 5) { if (c == 0) clip5 = (int)len; else clip3 = (int)len; if (hard) op = 5; }
                        o.num(len); o.ch(op < 9 ? OPS[op] : '?');
                    }
                }
                o.ch('\t');
                if (has_mate && mrid >= 0) { if (mrid == rid || rid < 0) o.ch('='); else o.put(contigs[mrid].name); o.ch('\t'); o.num((long long)mpos + 1); }
                else if (has_mate && rid >= 0) { o.put("=\t"); o.num((long long)pos + 1); }                                 // the mate is unmapped: it sits here
                else o.put("*\t0");
                o.ch('\t'); o.num(has_mate && rid >= 0 && mrid >= 0 ? tlen : 0); o.ch('\t');
                // SEQ: as sequenced, or its reverse complement; the clips are in alignment (reference-strand) order
                size_t b = 0, e = l_seq;
                if (hard) { b = (size_t)clip5; e = l_seq - (size_t)clip3; if (b > e) { b = 0; e = l_seq; } }
                if (l_seq == 0) o.ch('*');
                else if (flag & 0x10) { for (size_t i = b; i < e; ++i) o.ch(comp(seq[l_seq - 1 - i])); }
                else o.s.append(seq + b, e - b);
                o.put("\t*");
                if (!(flag & 4)) {
                    o.put("\tNM:i:"); o.num(nm);
                    if (!md.empty()) { o.put("\tMD:Z:"); o.put(md); }
                    o.put("\tAS:i:"); o.num(as);
                    if (xs >= 0) { o.put("\tXS:i:"); o.num(xs); }
                    if (!xa.empty()) { o.put("\tXA:Z:"); o.put(xa); }
                }
                o.ch('\n');
            }
        }
        if (p != end) return 0;
        char* res = (char*)malloc(o.s.size() + 1);
        if (!res) return 0;
        memcpy(res, o.s.data(), o.s.size() + 1);
        if (pBytes) *pBytes = o.s.size();
        return res;
    } catch (...) { return 0; }
}

}  // extern "C"
