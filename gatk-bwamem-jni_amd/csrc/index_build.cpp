// index_build.cpp -- fasta -> <prefix>.{pac,ann,amb,bwt,sa} (host builder).
//
// Replaces upstream bwa_idx_build (bwtindex.c) as called from the reference at
// ...BwaMemIndex.c:59 (BwaMemIndex.java:218-230).  Byte-exact with the reference's fixture
// files src/test/resources/ref.fa.{amb,ann,bwt,pac,sa} (tests/test_index.py): ambiguous bases
// are replaced by lrand48()&3 after srand48(11) exactly as upstream's packer does, the text is
// forward + reverse complement, the BWT drops the sentinel and records its rank as "primary",
// occ checkpoints are interleaved every 128 symbols, the SA is sampled every 32 ranks.
// Suffix sorting here is a host prefix-doubling sort meant for test-sized genomes; the
// GRCh38-scale path is the device builder in index_build_gpu.
#include "index_io.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <algorithm>
#include <numeric>

static inline int nt4(int c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

static bool write_file(const std::string& fn, const void* p, size_t n)
{
    FILE* fp = fopen(fn.c_str(), "wb");
    if (!fp) return false;
    bool ok = fwrite(p, 1, n, fp) == n;
    return fclose(fp) == 0 && ok;
}

// suffix array of t[0..n) (symbols 0..3) with an implicit unique smallest sentinel at n
static void suffix_sort(const std::vector<uint8_t>& t, std::vector<int64_t>& sa)
{
    const int64_t n = (int64_t)t.size();
    sa.resize(n);
    std::vector<int64_t> rank(n), tmp(n);
    std::iota(sa.begin(), sa.end(), 0);
    // initial rank: 12-mers (shorter at the end sort first via the length tie-break below)
    const int K = 12;
    auto key = [&](int64_t i) { int64_t k = 0; for (int j = 0; j < K; ++j) k = k * 5 + (i + j < n ? t[i + j] + 1 : 0); return k; };
    std::vector<int64_t> k0(n);
    for (int64_t i = 0; i < n; ++i) k0[i] = key(i);
    std::sort(sa.begin(), sa.end(), [&](int64_t a, int64_t b) { return k0[a] < k0[b]; });
    rank[sa[0]] = 0;
    for (int64_t i = 1; i < n; ++i) rank[sa[i]] = rank[sa[i - 1]] + (k0[sa[i]] != k0[sa[i - 1]]);
    for (int64_t h = K; rank[sa[n - 1]] != n - 1; h <<= 1) {
        auto r2 = [&](int64_t i) { return i + h < n ? rank[i + h] : -1; };
        std::sort(sa.begin(), sa.end(), [&](int64_t a, int64_t b) { return rank[a] != rank[b] ? rank[a] < rank[b] : r2(a) < r2(b); });
        tmp[sa[0]] = 0;
        for (int64_t i = 1; i < n; ++i)
            tmp[sa[i]] = tmp[sa[i - 1]] + (rank[sa[i]] != rank[sa[i - 1]] || r2(sa[i]) != r2(sa[i - 1]));
        rank.swap(tmp);
    }
}

// BWT (sentinel dropped) + occ interleave + sampled SA from the suffix array of fwd+revcomp
void index_pieces_from_sa(const std::vector<uint8_t>& text, const std::vector<int64_t>& sa, IndexPieces& p)
{
    const uint64_t n = text.size();
    p.seq_len = n;
    uint64_t cnt[4] = {0, 0, 0, 0};
    for (uint8_t c : text) ++cnt[c];
    p.L2[0] = 0;
    for (int c = 0; c < 4; ++c) p.L2[c + 1] = p.L2[c] + cnt[c];
    // rank k (sentinel-inclusive): k = 0 is the empty suffix, k = i+1 is sa[i]
    std::vector<uint8_t> b(n);
    uint64_t w = 0;
    p.primary = 0;
    for (uint64_t k = 0; k <= n; ++k) {
        int64_t pos = k == 0 ? (int64_t)n : sa[k - 1];
        if (pos == 0) { p.primary = k; continue; }
        b[w++] = text[pos - 1];
    }
    const uint64_t n_occ = (n + 127) / 128 + 1;
    p.bwt.assign((n + 15) / 16 + n_occ * 8, 0);
    uint64_t c4[4] = {0, 0, 0, 0}, k = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (i % 128 == 0) { memcpy(&p.bwt[k], c4, 32); k += 8; }
        if (i % 16 == 0) ++k;
        p.bwt[k - 1] |= (uint32_t)b[i] << ((~i & 15) << 1);
        ++c4[b[i]];
    }
    memcpy(&p.bwt[k], c4, 32);
    p.sa_intv = 32;
    const uint64_t n_sa = (n + 32) / 32;
    p.sa.assign(n_sa, 0);
    for (uint64_t j = 0; j < n_sa; ++j) {
        uint64_t r = j * 32;
        p.sa[j] = r == 0 ? (uint64_t)-1 : (uint64_t)sa[r - 1];
    }
}

struct FastaPack { IndexPieces p; std::vector<uint8_t> fwd; };

// upstream bns_fasta2bntseq + add1 (forward strand); names/comments follow kseq's split at the first blank
static bool pack_fasta(const std::string& fasta, FastaPack& out, std::string* err)
{
    FILE* fp = fopen(fasta.c_str(), "rb");
    if (!fp) { if (err) *err = "cannot open " + fasta; return false; }
    IndexPieces& p = out.p;
    p.seed = 11;
    srand48(p.seed);
    std::vector<uint8_t>& fwd = out.fwd;
    int c, lasts = 0;
    bool in_seq = false;
    ContigInfo* cur = nullptr;
    c = fgetc(fp);
    while (c != EOF) {
        if (c == '>') {
            std::string hdr;
            while ((c = fgetc(fp)) != EOF && c != '\n') hdr.push_back((char)c);
            while (!hdr.empty() && (hdr.back() == '\r')) hdr.pop_back();
            size_t sp = hdr.find_first_of(" \t");
            ContigInfo ci;
            ci.name = hdr.substr(0, sp);
            std::string comment;
            if (sp != std::string::npos) { size_t s2 = hdr.find_first_not_of(" \t", sp); if (s2 != std::string::npos) comment = hdr.substr(s2); }
            ci.anno = comment.empty() ? "(null)" : comment;      // upstream writes the literal "(null)"
            ci.gi = 0; ci.len = 0; ci.n_ambs = 0; ci.is_alt = 0;
            ci.offset = p.contigs.empty() ? 0 : p.contigs.back().offset + p.contigs.back().len;
            p.contigs.push_back(ci);
            cur = &p.contigs.back();
            lasts = 0; in_seq = true;
            if (c == '\n') c = fgetc(fp);
            continue;
        }
        if (in_seq && isgraph(c)) {
            int b = nt4(c);
            if (b >= 4) {
                if (lasts == c) ++p.holes.back().len;
                else {
                    HoleInfo h; h.len = 1; h.offset = cur->offset + cur->len; h.amb = (char)c;
                    p.holes.push_back(h);
                    ++cur->n_ambs;
                }
                b = (int)(lrand48() & 3);
            }
            lasts = c;
            fwd.push_back((uint8_t)b);
            ++cur->len;
        }
        c = fgetc(fp);
    }
    fclose(fp);
    p.l_pac = (int64_t)fwd.size();
    p.pac.assign((size_t)(p.l_pac / 4 + 1), 0);
    for (int64_t i = 0; i < p.l_pac; ++i) p.pac[i >> 2] |= (uint8_t)(fwd[i] << ((~i & 3) << 1));
    if (p.contigs.empty() || p.l_pac == 0) { if (err) *err = "no sequence in " + fasta; return false; }
    return true;
}

bool write_index_files(const IndexPieces& p, const std::string& prefix, std::string* err)
{
    auto fail = [&](const std::string& m) { if (err) *err = m; return false; };
    {   // .pac: packed bytes, then a zero byte if l_pac%4==0, then l_pac%4 (App. A.1)
        std::vector<uint8_t> f(p.pac.begin(), p.pac.begin() + ((p.l_pac >> 2) + ((p.l_pac & 3) == 0 ? 0 : 1)));
        if (p.l_pac % 4 == 0) f.push_back(0);
        f.push_back((uint8_t)(p.l_pac % 4));
        if (!write_file(prefix + ".pac", f.data(), f.size())) return fail("cannot write .pac");
    }
    {
        FILE* fp = fopen((prefix + ".ann").c_str(), "w");
        if (!fp) return fail("cannot write .ann");
        fprintf(fp, "%lld %d %u\n", (long long)p.l_pac, (int)p.contigs.size(), p.seed);
        for (const ContigInfo& c : p.contigs) {
            fprintf(fp, "%d %s", (int)c.gi, c.name.c_str());
            if (!c.anno.empty()) fprintf(fp, " %s\n", c.anno.c_str()); else fprintf(fp, "\n");
            fprintf(fp, "%lld %d %d\n", (long long)c.offset, c.len, c.n_ambs);
        }
        fclose(fp);
        fp = fopen((prefix + ".amb").c_str(), "w");
        if (!fp) return fail("cannot write .amb");
        fprintf(fp, "%lld %d %u\n", (long long)p.l_pac, (int)p.contigs.size(), (unsigned)p.holes.size());
        for (const HoleInfo& h : p.holes) fprintf(fp, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
        fclose(fp);
    }
    {
        std::vector<uint8_t> f(40 + p.bwt.size() * 4);
        memcpy(f.data(), &p.primary, 8);
        memcpy(f.data() + 8, &p.L2[1], 32);
        memcpy(f.data() + 40, p.bwt.data(), p.bwt.size() * 4);
        if (!write_file(prefix + ".bwt", f.data(), f.size())) return fail("cannot write .bwt");
    }
    {
        std::vector<uint8_t> f(56 + (p.sa.size() - 1) * 8);
        uint64_t intv = (uint64_t)p.sa_intv;
        memcpy(f.data(), &p.primary, 8);
        memcpy(f.data() + 8, &p.L2[1], 32);
        memcpy(f.data() + 40, &intv, 8);
        memcpy(f.data() + 48, &p.seq_len, 8);
        memcpy(f.data() + 56, p.sa.data() + 1, (p.sa.size() - 1) * 8);
        if (!write_file(prefix + ".sa", f.data(), f.size())) return fail("cannot write .sa");
    }
    return true;
}

bool build_index_files(const std::string& fasta, const std::string& prefix, std::string* err)
{
    FastaPack fp;
    if (!pack_fasta(fasta, fp, err)) return false;
    const int64_t l = fp.p.l_pac;
    const char* eb = getenv("BWAMEM_HIP_INDEX_BUILDER");
    const bool force_host = eb && !strcmp(eb, "host"), force_dev = eb && !strcmp(eb, "device");
    if (!force_host && device_index_available()) {
        std::string derr;
        if (device_index_pieces(fp.fwd, fp.p, &derr)) return write_index_files(fp.p, prefix, err);
        if (force_dev) { if (err) *err = "device index builder: " + derr; return false; }
        fprintf(stderr, "[bwamem_hip] device index builder failed (%s): building on the host\n", derr.c_str());
    } else if (force_dev) { if (err) *err = "BWAMEM_HIP_INDEX_BUILDER=device but no HIP device is visible"; return false; }
    std::vector<uint8_t> text((size_t)(2 * l));
    for (int64_t i = 0; i < l; ++i) { text[i] = fp.fwd[i]; text[2 * l - 1 - i] = (uint8_t)(3 - fp.fwd[i]); }
    std::vector<int64_t> sa;
    suffix_sort(text, sa);
    index_pieces_from_sa(text, sa, fp.p);
    return write_index_files(fp.p, prefix, err);
}
