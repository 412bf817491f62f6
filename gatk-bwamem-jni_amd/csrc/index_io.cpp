// index_io.cpp -- index image parser / assembler and the five-file loader (host only).
//
// Replaces upstream bwa.c bwa_idx_load / bwa_idx2mem / bwa_mem2idx, bwt.c bwt_restore_bwt /
// bwt_restore_sa and bntseq.c bns_restore as reached from the reference at jnibwa.c:127-128,160.
// The image is a concatenation of x86-64 struct dumps whose pointer fields are stale
// (SURVEY.md App. A.4); they are written as zero here and ignored on read.
#include "index_io.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <fstream>
#include <sstream>

namespace {

// byte offsets inside the dumped structs (App. A.4)
enum {
    BWT_HDR = 1120, BWT_PRIMARY = 0, BWT_L2 = 8, BWT_SEQLEN = 48, BWT_SIZE = 56, BWT_CNT = 72, BWT_SAINTV = 1096, BWT_NSA = 1104,
    BNS_HDR = 48, BNS_LPAC = 0, BNS_NSEQS = 8, BNS_SEED = 12, BNS_NHOLES = 24,
    ANN_REC = 40, ANN_OFFSET = 0, ANN_LEN = 8, ANN_NAMBS = 12, ANN_GI = 16, ANN_ISALT = 20,
    AMB_REC = 16, AMB_OFFSET = 0, AMB_LEN = 8, AMB_CHAR = 12
};

template <typename T> T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return v; }
template <typename T> void wr(uint8_t* p, T v) { memcpy(p, &v, sizeof(T)); }

bool read_file(const std::string& fn, std::vector<uint8_t>& out)
{
    FILE* fp = fopen(fn.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, fp) == (size_t)n;
    fclose(fp);
    return ok;
}

} // namespace

bool parse_index_image(const uint8_t* mem, size_t l_mem, HostIndex& ix)
{
    size_t k = 0;
    if (l_mem < BWT_HDR) return false;
    ix.mem = mem; ix.l_mem = l_mem;
    ix.primary = rd<uint64_t>(mem + BWT_PRIMARY);
    for (int i = 0; i < 5; ++i) ix.L2[i] = rd<uint64_t>(mem + BWT_L2 + 8 * i);
    ix.seq_len = rd<uint64_t>(mem + BWT_SEQLEN);
    ix.bwt_size = rd<uint64_t>(mem + BWT_SIZE);
    ix.sa_intv = rd<int32_t>(mem + BWT_SAINTV);
    ix.n_sa = rd<uint64_t>(mem + BWT_NSA);
    k = BWT_HDR;
    if (ix.sa_intv <= 0 || (ix.sa_intv & (ix.sa_intv - 1))) return false;
    if (ix.bwt_size > (l_mem - k) / 4) return false;
    ix.bwt = (const uint32_t*)(mem + k); k += ix.bwt_size * 4;
    if (ix.n_sa > (l_mem - k) / 8) return false;
    ix.sa = (const uint64_t*)(mem + k); k += ix.n_sa * 8;
    if (k + BNS_HDR > l_mem) return false;
    const uint8_t* b = mem + k;
    ix.l_pac = rd<int64_t>(b + BNS_LPAC);
    int32_t n_seqs = rd<int32_t>(b + BNS_NSEQS), n_holes = rd<int32_t>(b + BNS_NHOLES);
    ix.seed = rd<uint32_t>(b + BNS_SEED);
    k += BNS_HDR;
    if (n_seqs < 0 || n_holes < 0 || ix.l_pac < 0) return false;
    if ((size_t)n_holes > (l_mem - k) / AMB_REC) return false;
    ix.holes.resize(n_holes);
    for (int i = 0; i < n_holes; ++i, k += AMB_REC) {
        ix.holes[i].offset = rd<int64_t>(mem + k + AMB_OFFSET);
        ix.holes[i].len = rd<int32_t>(mem + k + AMB_LEN);
        ix.holes[i].amb = (char)mem[k + AMB_CHAR];
    }
    if ((size_t)n_seqs > (l_mem - k) / ANN_REC) return false;
    ix.contigs.resize(n_seqs);
    for (int i = 0; i < n_seqs; ++i, k += ANN_REC) {
        ContigInfo& c = ix.contigs[i];
        c.offset = rd<int64_t>(mem + k + ANN_OFFSET);
        c.len = rd<int32_t>(mem + k + ANN_LEN);
        c.n_ambs = rd<int32_t>(mem + k + ANN_NAMBS);
        c.gi = rd<uint32_t>(mem + k + ANN_GI);
        c.is_alt = rd<int32_t>(mem + k + ANN_ISALT);
    }
    for (int i = 0; i < n_seqs; ++i) {       // name\0anno\0 per contig
        for (int part = 0; part < 2; ++part) {
            const void* e = memchr(mem + k, 0, l_mem - k);
            if (!e) return false;
            std::string s((const char*)(mem + k), (const char*)e);
            (part ? ix.contigs[i].anno : ix.contigs[i].name) = s;
            k = (size_t)((const uint8_t*)e - mem) + 1;
        }
    }
    ix.pac = mem + k; k += (size_t)(ix.l_pac / 4 + 1);
    return k == l_mem;                       // upstream asserts exactly this
}

std::vector<uint8_t> image_from_pieces(const IndexPieces& p)
{
    size_t strbytes = 0;
    for (const ContigInfo& c : p.contigs) strbytes += c.name.size() + c.anno.size() + 2;
    size_t total = BWT_HDR + p.bwt.size() * 4 + p.sa.size() * 8 + BNS_HDR + p.holes.size() * AMB_REC
                 + p.contigs.size() * ANN_REC + strbytes + (size_t)(p.l_pac / 4 + 1);
    std::vector<uint8_t> img(total, 0);
    uint8_t* m = img.data();
    wr<uint64_t>(m + BWT_PRIMARY, p.primary);
    for (int i = 0; i < 5; ++i) wr<uint64_t>(m + BWT_L2 + 8 * i, p.L2[i]);
    wr<uint64_t>(m + BWT_SEQLEN, p.seq_len);
    wr<uint64_t>(m + BWT_SIZE, (uint64_t)p.bwt.size());
    for (int i = 0; i < 256; ++i) {          // the byte -> 4 packed counts lookup upstream keeps in the header
        uint32_t x = 0;
        for (int j = 0; j < 4; ++j)
            x |= (uint32_t)(((i & 3) == j) + ((i >> 2 & 3) == j) + ((i >> 4 & 3) == j) + (i >> 6 == j)) << (j << 3);
        wr<uint32_t>(m + BWT_CNT + 4 * i, x);
    }
    wr<int32_t>(m + BWT_SAINTV, p.sa_intv);
    wr<uint64_t>(m + BWT_NSA, (uint64_t)p.sa.size());
    size_t k = BWT_HDR;
    memcpy(m + k, p.bwt.data(), p.bwt.size() * 4); k += p.bwt.size() * 4;
    memcpy(m + k, p.sa.data(), p.sa.size() * 8); k += p.sa.size() * 8;
    wr<int64_t>(m + k + BNS_LPAC, p.l_pac);
    wr<int32_t>(m + k + BNS_NSEQS, (int32_t)p.contigs.size());
    wr<uint32_t>(m + k + BNS_SEED, p.seed);
    wr<int32_t>(m + k + BNS_NHOLES, (int32_t)p.holes.size());
    k += BNS_HDR;
    for (const HoleInfo& h : p.holes) {
        wr<int64_t>(m + k + AMB_OFFSET, h.offset); wr<int32_t>(m + k + AMB_LEN, h.len); m[k + AMB_CHAR] = (uint8_t)h.amb;
        k += AMB_REC;
    }
    for (const ContigInfo& c : p.contigs) {
        wr<int64_t>(m + k + ANN_OFFSET, c.offset); wr<int32_t>(m + k + ANN_LEN, c.len); wr<int32_t>(m + k + ANN_NAMBS, c.n_ambs);
        wr<uint32_t>(m + k + ANN_GI, c.gi); wr<int32_t>(m + k + ANN_ISALT, c.is_alt);
        k += ANN_REC;
    }
    for (const ContigInfo& c : p.contigs) {
        memcpy(m + k, c.name.c_str(), c.name.size() + 1); k += c.name.size() + 1;
        memcpy(m + k, c.anno.c_str(), c.anno.size() + 1); k += c.anno.size() + 1;
    }
    memcpy(m + k, p.pac.data(), (size_t)(p.l_pac / 4 + 1));
    return img;
}

std::vector<uint8_t> image_from_index_files(const std::string& prefix, std::string* err)
{
    auto fail = [&](const std::string& msg) { if (err) *err = msg; return std::vector<uint8_t>(); };
    IndexPieces p;
    std::vector<uint8_t> f;
    // .bwt: u64 primary, u64 L2[1..4], then the interleaved words (App. A.2)
    if (!read_file(prefix + ".bwt", f) || f.size() < 40 || (f.size() - 40) % 4) return fail("cannot read " + prefix + ".bwt");
    p.primary = rd<uint64_t>(f.data());
    for (int i = 1; i < 5; ++i) p.L2[i] = rd<uint64_t>(f.data() + 8 * i);
    p.seq_len = p.L2[4];
    p.bwt.resize((f.size() - 40) / 4);
    memcpy(p.bwt.data(), f.data() + 40, f.size() - 40);
    // .sa: u64 primary, u64 L2[1..4], u64 sa_intv, u64 seq_len, then sa[1..n_sa) (App. A.3)
    if (!read_file(prefix + ".sa", f) || f.size() < 56) return fail("cannot read " + prefix + ".sa");
    if (rd<uint64_t>(f.data()) != p.primary || rd<uint64_t>(f.data() + 48) != p.seq_len) return fail("SA-BWT inconsistency");
    uint64_t sa_intv = rd<uint64_t>(f.data() + 40);
    if (sa_intv == 0 || sa_intv > (1u << 20)) return fail("bad sa_intv");
    p.sa_intv = (int)sa_intv;
    uint64_t n_sa = (p.seq_len + sa_intv) / sa_intv;
    if (f.size() < 56 + (n_sa - 1) * 8) return fail("truncated " + prefix + ".sa");
    p.sa.resize(n_sa);
    p.sa[0] = (uint64_t)-1;
    memcpy(p.sa.data() + 1, f.data() + 56, (n_sa - 1) * 8);
    // .ann
    {
        std::ifstream in(prefix + ".ann");
        if (!in) return fail("cannot read " + prefix + ".ann");
        std::string line;
        long long l_pac; int n_seqs; unsigned seed;
        if (!std::getline(in, line) || sscanf(line.c_str(), "%lld %d %u", &l_pac, &n_seqs, &seed) != 3) return fail("bad .ann header");
        p.l_pac = l_pac; p.seed = seed;
        p.contigs.resize(n_seqs);
        for (int i = 0; i < n_seqs; ++i) {
            ContigInfo& c = p.contigs[i];
            if (!std::getline(in, line)) return fail("truncated .ann");
            std::istringstream ls(line);
            unsigned gi; std::string name;
            if (!(ls >> gi >> name)) return fail("bad .ann record");
            std::string rest;
            std::getline(ls, rest);                       // " comment" (leading blank kept by getline)
            c.gi = gi; c.name = name;
            c.anno = (rest.size() > 1 && rest != " (null)") ? rest.substr(1) : std::string();
            long long off; int len, n_ambs;
            if (!std::getline(in, line) || sscanf(line.c_str(), "%lld %d %d", &off, &len, &n_ambs) != 3) return fail("bad .ann record");
            c.offset = off; c.len = len; c.n_ambs = n_ambs; c.is_alt = 0;
        }
    }
    // .amb
    {
        std::ifstream in(prefix + ".amb");
        if (!in) return fail("cannot read " + prefix + ".amb");
        long long l_pac; int n_seqs, n_holes;
        if (!(in >> l_pac >> n_seqs >> n_holes) || l_pac != p.l_pac || n_seqs != (int)p.contigs.size()) return fail("inconsistent .ann and .amb");
        p.holes.resize(n_holes);
        for (int i = 0; i < n_holes; ++i) {
            long long off; int len; std::string c;
            if (!(in >> off >> len >> c)) return fail("truncated .amb");
            p.holes[i].offset = off; p.holes[i].len = len; p.holes[i].amb = c[0];
        }
    }
    // optional .alt: first column of every non-'@' line names an ALT contig
    {
        std::ifstream in(prefix + ".alt");
        std::string line;
        while (in && std::getline(in, line)) {
            if (line.empty() || line[0] == '@') continue;
            std::string name = line.substr(0, line.find_first_of("\t\r"));
            for (ContigInfo& c : p.contigs) if (c.name == name) c.is_alt = 1;
        }
    }
    // .pac: the first l_pac/4+1 bytes are the packed forward strand (App. A.1)
    if (!read_file(prefix + ".pac", f) || f.size() < (size_t)(p.l_pac / 4 + 1)) return fail("cannot read " + prefix + ".pac");
    p.pac.assign(f.begin(), f.begin() + (p.l_pac / 4 + 1));
    return image_from_pieces(p);
}
