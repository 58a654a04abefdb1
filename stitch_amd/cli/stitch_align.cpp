// stitch-align: command-line front end of libstitch_amd.so with the flags of `stitch align`
// (fg-stitch-cli/src/commands/align.rs:94-275).  FASTA/FASTQ(.gz) in, SAM text or BAM (BGZF) out on stdout.
//
//   reference FASTA   -> stitch_index_build            (util/target_seq.rs:69-123: name = first word of the header)
//   reads             -> batches of stitch_align_batch (align/io.rs: FASTQ 4-line records, FASTA multi-line)
//   per read          -> stitch_format_sam             (aligners/mod.rs:622-973)
//   header            -> @HD, @SQ per target, @PG      (align.rs:393-416)
//
// `--devices 0,1,...` runs one worker PROCESS per GPU (the counterpart of the reference's worker threads, align.rs:345-390): the
// parent — which never touches a GPU — builds the reference index once and hands it to the workers as the serialized blob,
// scans the read file ONCE without keeping it (per record: byte offset, length and a 64-bit hash of the bases), cuts the stream
// into contiguous blocks at read-group boundaries (stitch_shard_range on the hashes), starts the workers (fork + exec of this
// program, before any GPU call; each worker seeks to the byte offset of its block) and concatenates their records in rank order
// behind the header: byte for byte what one device writes.  With `--index-via rccl` only rank 0 reads the blob and the others
// receive it through ONE native ncclBroadcast (RCCL over xGMI, librccl.so loaded at run time): the collective BASELINE.json's
// north_star names; the default hands the blob over as a file.
// The alignment itself only exists on the GPU: without a device the program stops with the library's error.
// `--dry-run` parses the inputs and writes the header only (no device needed); `--convert-sam FILE` re-encodes a SAM
// file as BAM (what `--output-format bam` does to the records it produces) so that the encoder can be tested alone.
#include <dlfcn.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <hip/hip_runtime_api.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/stitch_gpu.h"

namespace {

[[noreturn]] void die(const std::string& m) { fprintf(stderr, "stitch-align: %s\n", m.c_str()); exit(2); }

// ---- FASTA / FASTQ ----------------------------------------------------------------------------------------------
struct Rec { std::string head, seq, qual; bool has_qual = false; };

struct LineReader {       // plain or gzip (zlib reads both transparently)
    gzFile f = nullptr; std::string path;
    explicit LineReader(const std::string& p) : path(p) { f = p == "-" ? gzdopen(0, "rb") : gzopen(p.c_str(), "rb"); if (!f) die("cannot open " + p); gzbuffer(f, 1 << 17); }
    ~LineReader() { if (f) gzclose(f); }
    bool line(std::string& out) {
        out.clear();
        char buf[1 << 16];
        for (;;) {
            if (!gzgets(f, buf, sizeof buf)) return !out.empty();
            size_t n = strlen(buf);
            const bool eol = n && buf[n - 1] == '\n';
            if (eol) --n;
            if (n && buf[n - 1] == '\r') --n;
            out.append(buf, n);
            if (eol) return true;
        }
    }
};

struct FastxReader {
    LineReader in; bool fastq; std::string pending; bool have_pending = false;
    long long rec_off = 0, pending_off = 0;      // byte offset (in the uncompressed stream) of the header line of the record just returned
    FastxReader(const std::string& p, bool fq) : in(p), fastq(fq) {}
    // continue at a record boundary found by an earlier scan (plain files: a seek; gzip: zlib inflates up to there, without parsing)
    void seek(long long off) { if (off > 0 && gzseek(in.f, (z_off_t)off, SEEK_SET) < 0) die("cannot seek in " + in.path); have_pending = false; }
    bool next(Rec& r) {
        std::string l;
        if (fastq) {
            do { rec_off = (long long)gztell(in.f); if (!in.line(l)) return false; } while (l.empty());
            if (l[0] != '@') die("malformed FASTQ record in " + in.path + ": " + l.substr(0, 40));
            r.head = l.substr(1); r.has_qual = true;
            if (!in.line(r.seq) || !in.line(l) || l.empty() || l[0] != '+' || !in.line(r.qual)) die("truncated FASTQ record in " + in.path);
            if (r.qual.size() != r.seq.size()) die("FASTQ record with unequal sequence and quality lengths: " + r.head);
            return true;
        }
        if (!have_pending) { do { pending_off = (long long)gztell(in.f); if (!in.line(l)) return false; } while (l.empty()); pending = l; have_pending = true; }
        if (pending[0] != '>') die("malformed FASTA record in " + in.path);
        r.head = pending.substr(1); r.seq.clear(); r.qual.clear(); r.has_qual = false; have_pending = false; rec_off = pending_off;
        for (;;) { const long long at = (long long)gztell(in.f); if (!in.line(l)) break; if (!l.empty() && l[0] == '>') { pending = l; pending_off = at; have_pending = true; break; } r.seq += l; }
        return true;
    }
};

std::string first_word(const std::string& s) {
    size_t a = 0; while (a < s.size() && isspace((unsigned char)s[a])) ++a;
    size_t b = a; while (b < s.size() && !isspace((unsigned char)s[b])) ++b;
    return s.substr(a, b - a);
}
bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

// ---- BGZF / BAM -------------------------------------------------------------------------------------------------
struct Out {
    bool bam = false; int level = 0; std::string blk;
    void raw(const void* p, size_t n) { if (fwrite(p, 1, n, stdout) != n) die("write failed"); }
    void put(const void* p, size_t n) {
        if (!bam) { raw(p, n); return; }
        const char* c = (const char*)p;
        while (n) { size_t k = std::min(n, (size_t)0xFF00 - blk.size()); blk.append(c, k); c += k; n -= k; if (blk.size() >= 0xFF00) flush_block(); }
    }
    void flush_block() {
        if (!bam || blk.empty()) return;
        write_block(blk.data(), blk.size()); blk.clear();
    }
    void write_block(const char* data, size_t len) {
        std::vector<uint8_t> cmp(len + 1024);
        z_stream zs{}; if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2");
        zs.next_in = (Bytef*)data; zs.avail_in = (uInt)len; zs.next_out = cmp.data(); zs.avail_out = (uInt)cmp.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("deflate");
        const size_t clen = zs.total_out; deflateEnd(&zs);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)data, (uInt)len);
        const uint16_t bsize = (uint16_t)(clen + 25);             // total block size - 1
        uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xFF), (uint8_t)(bsize >> 8)};
        raw(hdr, 18); raw(cmp.data(), clen);
        uint8_t tail[8]; for (int k = 0; k < 4; ++k) { tail[k] = (uint8_t)(crc >> (8 * k)); tail[4 + k] = (uint8_t)((uint32_t)len >> (8 * k)); }
        raw(tail, 8);
    }
    void finish() {
        if (bam) { flush_block(); static const uint8_t eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0}; raw(eof, 28); }
        fflush(stdout);
    }
};

void le32(std::string& b, uint32_t v) { for (int k = 0; k < 4; ++k) b.push_back((char)(v >> (8 * k))); }
void le16(std::string& b, uint16_t v) { b.push_back((char)(v & 0xFF)); b.push_back((char)(v >> 8)); }

int reg2bin(int64_t beg, int64_t end) {      // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct BamEncoder {
    std::map<std::string, int> ref_id; std::vector<std::pair<std::string, uint32_t>> refs;
    void header(Out& o, const std::string& text) {
        std::string b = "BAM\1"; le32(b, (uint32_t)text.size()); b += text; le32(b, (uint32_t)refs.size());
        for (auto& r : refs) { le32(b, (uint32_t)r.first.size() + 1); b += r.first; b.push_back('\0'); le32(b, r.second); }
        o.put(b.data(), b.size());
    }
    int rid(const std::string& n) const { if (n == "*") return -1; auto it = ref_id.find(n); if (it == ref_id.end()) die("record refers to unknown reference " + n); return it->second; }
    // one SAM text line -> one BAM record.  Integer tags are written as int32 ('i'); the reference's noodles encoder
    // is not pinned by any test, SAM text is the level at which outputs are compared.
    void record(Out& o, const std::string& line) {
        std::vector<std::string> f; size_t a = 0;
        for (;;) { size_t t = line.find('\t', a); f.push_back(line.substr(a, t == std::string::npos ? t : t - a)); if (t == std::string::npos) break; a = t + 1; }
        if (f.size() < 11) die("malformed SAM record: " + line.substr(0, 60));
        const std::string& qname = f[0]; const uint32_t flag = (uint32_t)strtoul(f[1].c_str(), nullptr, 10);
        const int ref = rid(f[2]); const int64_t pos = strtoll(f[3].c_str(), nullptr, 10) - 1; const uint32_t mapq = (uint32_t)strtoul(f[4].c_str(), nullptr, 10);
        std::vector<uint32_t> cig; int64_t reflen = 0;
        if (f[5] != "*") {
            const char* p = f[5].c_str();
            while (*p) { char* e; unsigned long n = strtoul(p, &e, 10); const char* ops = "MIDNSHP=X"; const char* q = strchr(ops, *e); if (!q || !*e) die("bad CIGAR " + f[5]);
                const uint32_t op = (uint32_t)(q - ops); cig.push_back((uint32_t)n << 4 | op); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += (int64_t)n; p = e + 1; }
        }
        const int nref = f[6] == "=" ? ref : rid(f[6]); const int64_t npos = strtoll(f[7].c_str(), nullptr, 10) - 1; const int32_t tlen = (int32_t)strtol(f[8].c_str(), nullptr, 10);
        const std::string seq = f[9] == "*" ? std::string() : f[9];
        std::string b;
        le32(b, (uint32_t)ref); le32(b, (uint32_t)pos);
        b.push_back((char)(qname.size() + 1)); b.push_back((char)mapq); le16(b, (uint16_t)reg2bin(pos < 0 ? 0 : pos, (pos < 0 ? 0 : pos) + (reflen ? reflen : 1)));
        le16(b, (uint16_t)cig.size()); le16(b, (uint16_t)flag); le32(b, (uint32_t)seq.size());
        le32(b, (uint32_t)nref); le32(b, (uint32_t)npos); le32(b, (uint32_t)tlen);
        b += qname; b.push_back('\0');
        for (uint32_t c : cig) le32(b, c);
        static const char* codes = "=ACMGRSVTWYHKDBN";
        for (size_t k = 0; k < seq.size(); k += 2) {
            auto code = [&](char c) -> uint8_t { const char* q = strchr(codes, toupper((unsigned char)c)); return q && c ? (uint8_t)(q - codes) : 15; };
            b.push_back((char)(code(seq[k]) << 4 | (k + 1 < seq.size() ? code(seq[k + 1]) : 0)));
        }
        if (f[10] == "*") b.append(seq.size(), (char)0xFF); else { if (f[10].size() != seq.size()) die("SEQ/QUAL length mismatch in " + qname); for (char c : f[10]) b.push_back((char)(c - 33)); }
        for (size_t k = 11; k < f.size(); ++k) {
            const std::string& t = f[k]; if (t.size() < 5 || t[2] != ':' || t[4] != ':') die("bad tag " + t);
            b.push_back(t[0]); b.push_back(t[1]); const std::string v = t.substr(5);
            switch (t[3]) {
                case 'i': b.push_back('i'); le32(b, (uint32_t)strtol(v.c_str(), nullptr, 10)); break;
                case 'A': b.push_back('A'); b.push_back(v.empty() ? ' ' : v[0]); break;
                case 'f': { b.push_back('f'); float x = strtof(v.c_str(), nullptr); uint32_t u; memcpy(&u, &x, 4); le32(b, u); break; }
                case 'Z': b.push_back('Z'); b += v; b.push_back('\0'); break;
                default: die("unsupported tag type in " + t);
            }
        }
        std::string sz; le32(sz, (uint32_t)b.size()); o.put(sz.data(), 4); o.put(b.data(), b.size());
    }
};

// ---- arguments --------------------------------------------------------------------------------------------------
struct Args {
    std::string reads_fastq, reads_fasta, ref_fasta, out_format = "bam", convert_sam;      // (the reference writes BAM to stdout, commands/align.rs:393-416)
    stitch_opts o; int jump_score = -10; bool js_same = false, js_opp = false, js_inter = false;
    int threads = 2, compression = 0, device = 0; uint32_t batch = 1024; bool decompress = false, dry_run = false;
    std::vector<int> devices;                       // --devices: one worker process per entry
    // worker mode (set by the parent of a --devices run): records [shard_lo, shard_hi) only, SAM records without header to
    // shard_out, the reference index from the serialized blob
    long shard_lo = -1, shard_hi = -1; std::string shard_out, index_blob;
    long long shard_offset = 0;                     // byte offset of record shard_lo in the (uncompressed) read stream
    std::string index_via = "file";                 // "file" | "rccl": how the workers of a --devices run get the index blob
    int rccl_rank = -1, rccl_world = 0; std::string rccl_id;      // worker side of --index-via rccl
};

const char* USAGE =
    "Usage: stitch-align (-f READS.fastq | -a READS.fasta) -r REF.fasta [options] > out.sam\n"
    "  -f, --reads-fastq PATH      input FASTQ (plain or gzip)\n"
    "  -a, --reads-fasta PATH      input FASTA (plain or gzip)\n"
    "  -r, --ref-fasta PATH        reference vector/plasmid/construct FASTA\n"
    "  -d, --double-strand         align to both strands simultaneously\n"
    "  -t, --threads N             accepted for compatibility (the GPU does the work)\n"
    "  -z, --decompress            accepted for compatibility (gzip is detected)\n"
    "  -p, --pre-align  -k K  -w W  -s SCORE  -x BOOL   banded pre-alignment filter (local mode; k 12, w 50, score 100, subset true)\n"
    "  -S, --soft-clip             soft-clip secondary alignments too\n"
    "  -X, --use-eq-and-x          =/X CIGAR operators instead of M\n"
    "  -A/-B/-O/-E/-J N            match, mismatch, gap open, gap extend, jump scores (1 -4 -6 -2 -10)\n"
    "      --jump-score-same-contig-and-strand N  --jump-score-same-contig-opposite-strand N  --jump-score-inter-contig N\n"
    "  -m, --mode MODE             Local | QueryLocal | TargetLocal | Global\n"
    "  -P, --pick-primary HOW      QueryLength | Score\n"
    "  -C, --circular              treat the targets as circular;  --circular-slop N (20)\n"
    "      --filter-secondary      --filter-secondary-pct X (10)\n"
    "      --suboptimal            --suboptimal-pct X (20)\n"
    "  -c, --compression N         BGZF level of the BAM output (0)\n"
    "      --output-format FMT     bam (default, as the reference: BGZF level -c) | sam\n"
    "      --device N  --batch N   GPU ordinal (0), reads per library call (1024)\n"
    "      --devices A,B,...       one worker process per listed GPU; reads are cut into contiguous blocks at read-group\n"
    "                              boundaries, the output is the single-device output (a read FILE is needed, not stdin)\n"
    "      --index-via file|rccl   how the workers of a --devices run get the reference index: the serialized blob as a file\n"
    "                              (default), or one native RCCL broadcast from the first worker (one GPU per worker needed)\n"
    "      --dry-run               parse the inputs, write the header, align nothing (no GPU needed)\n";

bool parse_bool(const std::string& v, bool& out) {
    std::string s; for (char c : v) s.push_back((char)tolower((unsigned char)c));
    if (s == "true" || s == "1" || s == "yes") { out = true; return true; }
    if (s == "false" || s == "0" || s == "no") { out = false; return true; }
    return false;
}
std::string lower(std::string s) { for (auto& c : s) c = (char)tolower((unsigned char)c); return s; }

Args parse(int argc, char** argv) {
    Args a; stitch_opts_default(&a.o);
    auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]); return argv[++i]; };
    auto flag = [&](int& i) -> bool { bool v = true; if (i + 1 < argc && parse_bool(argv[i + 1], v)) ++i; return v; };     // `-x true` and bare `-x`
    auto num = [&](int& i) -> int { const std::string v = need(i); char* e; long x = strtol(v.c_str(), &e, 10); if (*e || v.empty()) die("not an integer: " + v); return (int)x; };
    auto real = [&](int& i) -> float { const std::string v = need(i); char* e; float x = strtof(v.c_str(), &e); if (*e || v.empty()) die("not a number: " + v); return x; };
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        const size_t eq = k.rfind("--", 0) == 0 ? k.find('=') : std::string::npos;
        std::vector<char*> spliced;
        if (eq != std::string::npos) die("use `--flag value`, not `--flag=value`: " + k);
        if (k == "-h" || k == "--help") { fputs(USAGE, stdout); exit(0); }
        else if (k == "-V" || k == "--version") { printf("stitch-align %s\n", stitch_version()); exit(0); }
        else if (k == "-f" || k == "--reads-fastq") a.reads_fastq = need(i);
        else if (k == "-a" || k == "--reads-fasta") a.reads_fasta = need(i);
        else if (k == "-r" || k == "--ref-fasta") a.ref_fasta = need(i);
        else if (k == "-d" || k == "--double-strand") a.o.double_strand = flag(i);
        else if (k == "-t" || k == "--threads") a.threads = num(i);
        else if (k == "-z" || k == "--decompress") a.decompress = flag(i);
        else if (k == "-p" || k == "--pre-align") a.o.pre_align = flag(i);
        else if (k == "-k" || k == "--k") a.o.kmer_size = num(i);
        else if (k == "-w" || k == "--w") a.o.band_width = num(i);
        else if (k == "-s" || k == "--pre-align-min-score") a.o.pre_align_min_score = num(i);
        else if (k == "-x" || k == "--pre-align-subset-contigs") a.o.pre_align_subset_contigs = flag(i);
        else if (k == "-S" || k == "--soft-clip") a.o.soft_clip = flag(i);
        else if (k == "-X" || k == "--use-eq-and-x") a.o.use_eq_and_x = flag(i);
        else if (k == "-A" || k == "--match-score") a.o.match_score = num(i);
        else if (k == "-B" || k == "--mismatch-score") a.o.mismatch_score = num(i);
        else if (k == "-O" || k == "--gap-open") a.o.gap_open = num(i);
        else if (k == "-E" || k == "--gap-extend") a.o.gap_extend = num(i);
        else if (k == "-J" || k == "--jump-score") a.jump_score = num(i);
        else if (k == "--jump-score-same-contig-and-strand") { a.o.jump_same = num(i); a.js_same = true; }
        else if (k == "--jump-score-same-contig-opposite-strand") { a.o.jump_opposite = num(i); a.js_opp = true; }
        else if (k == "--jump-score-inter-contig") { a.o.jump_inter = num(i); a.js_inter = true; }
        else if (k == "-m" || k == "--mode") {
            const std::string v = lower(need(i));
            if (v == "local") a.o.mode = 0; else if (v == "querylocal" || v == "query-local") a.o.mode = 1;
            else if (v == "targetlocal" || v == "target-local") a.o.mode = 2; else if (v == "global") a.o.mode = 3; else die("unknown mode " + v);
        }
        else if (k == "-P" || k == "--pick-primary") { const std::string v = lower(need(i)); if (v == "querylength" || v == "query-length") a.o.pick_primary = 0; else if (v == "score") a.o.pick_primary = 1; else die("unknown primary picking strategy " + v); }
        else if (k == "-C" || k == "--circular") a.o.circular = flag(i);
        else if (k == "--circular-slop") a.o.circular_slop = num(i);
        else if (k == "--filter-secondary") a.o.filter_secondary = flag(i);
        else if (k == "--filter-secondary-pct") a.o.filter_secondary_pct = real(i);
        else if (k == "--suboptimal") a.o.suboptimal = flag(i);
        else if (k == "--suboptimal-pct") a.o.suboptimal_pct = real(i);
        else if (k == "-c" || k == "--compression") a.compression = num(i);
        else if (k == "--output-format") a.out_format = lower(need(i));
        else if (k == "--device") a.device = num(i);
        else if (k == "--devices") { const std::string v = need(i); size_t p = 0; while (p <= v.size()) { size_t e = v.find(',', p); if (e == std::string::npos) e = v.size(); if (e > p) a.devices.push_back(atoi(v.substr(p, e - p).c_str())); p = e + 1; } if (a.devices.empty()) die("--devices needs a list of GPU ordinals"); }
        else if (k == "--shard") { a.shard_lo = num(i); a.shard_hi = num(i); }
        else if (k == "--shard-out") a.shard_out = need(i);
        else if (k == "--shard-offset") a.shard_offset = atoll(need(i).c_str());
        else if (k == "--index-via") { a.index_via = lower(need(i)); if (a.index_via != "file" && a.index_via != "rccl") die("--index-via file|rccl"); }
        else if (k == "--rccl") { a.rccl_rank = num(i); a.rccl_world = num(i); a.rccl_id = need(i); }
        else if (k == "--index-blob") a.index_blob = need(i);
        else if (k == "--batch") a.batch = (uint32_t)std::max(1, num(i));
        else if (k == "--dry-run") a.dry_run = true;
        else if (k == "--convert-sam") a.convert_sam = need(i);
        else die("unknown argument " + k + "\n" + USAGE);
    }
    // Options::contig_scoring (aligners/mod.rs:143-152): the three jump scores default to --jump-score
    if (!a.js_same) a.o.jump_same = a.jump_score;
    if (!a.js_opp) a.o.jump_opposite = a.jump_score;
    if (!a.js_inter) a.o.jump_inter = a.jump_score;
    if (a.out_format != "sam" && a.out_format != "bam") die("--output-format must be sam or bam");
    if (a.compression < 0 || a.compression > 9) die("--compression must be 0..9");
    return a;
}

std::string sam_header(const std::vector<std::pair<std::string, uint32_t>>& refs, int argc, char** argv) {
    std::string h = "@HD\tVN:1.6\n";
    for (auto& r : refs) h += "@SQ\tSN:" + r.first + "\tLN:" + std::to_string(r.second) + "\n";
    std::string cl; for (int i = 0; i < argc; ++i) { if (i) cl += ' '; cl += argv[i]; }
    h += std::string("@PG\tID:stitch\tPN:stitch\tVN:") + stitch_version() + "\tCL:" + cl + "\n";
    return h;
}

// ---- the one collective of a multi-GPU run: the serialized reference index, broadcast from rank 0 over RCCL ------------------------
// librccl.so is loaded at run time (the single-GPU path does not need it).  Rank 0 creates the unique id and leaves it in a file
// the parent named; the others wait for that file.  The blob's length goes first (8 bytes), then the blob.
struct NcclId { char internal[128]; };
std::string rccl_broadcast_blob(int rank, int world, const std::string& id_path, const std::string& blob_in) {
    if (!getenv("NCCL_SOCKET_IFNAME")) setenv("NCCL_SOCKET_IFNAME", "lo", 0);      // (one node: the bootstrap needs no other interface)
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) die(std::string("--index-via rccl: cannot load librccl.so: ") + dlerror());
    typedef int (*get_id_t)(NcclId*); typedef int (*init_t)(void**, int, NcclId, int); typedef int (*bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef int (*destroy_t)(void*); typedef const char* (*errstr_t)(int);
    auto get_id = (get_id_t)dlsym(h, "ncclGetUniqueId"); auto init = (init_t)dlsym(h, "ncclCommInitRank"); auto bcast = (bcast_t)dlsym(h, "ncclBroadcast");
    auto destroy = (destroy_t)dlsym(h, "ncclCommDestroy"); auto errstr = (errstr_t)dlsym(h, "ncclGetErrorString");
    if (!get_id || !init || !bcast || !destroy) die("--index-via rccl: librccl.so lacks the entry points");
    auto ok = [&](int rc, const char* what) { if (rc != 0) die(std::string("--index-via rccl: ") + what + ": " + (errstr ? errstr(rc) : "error")); };
    NcclId id{};
    if (rank == 0) {
        ok(get_id(&id), "ncclGetUniqueId");
        const std::string tmp = id_path + ".tmp"; FILE* f = fopen(tmp.c_str(), "wb"); if (!f || fwrite(&id, 1, sizeof id, f) != sizeof id) die("cannot write " + tmp); fclose(f);
        if (rename(tmp.c_str(), id_path.c_str()) != 0) die("cannot publish " + id_path);
    } else {
        for (int tries = 0;; ++tries) {            // (bounded: 60 s)
            FILE* f = fopen(id_path.c_str(), "rb");
            if (f) { const size_t n = fread(&id, 1, sizeof id, f); fclose(f); if (n == sizeof id) break; }
            if (tries > 6000) die("--index-via rccl: rank 0 never published the unique id");
            usleep(10000);
        }
    }
    void* comm = nullptr; ok(init(&comm, world, id, rank), "ncclCommInitRank");
    unsigned long long len = rank == 0 ? blob_in.size() : 0; unsigned long long* d_len = nullptr; uint8_t* d_blob = nullptr;
    if (hipMalloc((void**)&d_len, 8) != hipSuccess) die("hipMalloc failed");
    if (hipMemcpy(d_len, &len, 8, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy failed");
    ok(bcast(d_len, d_len, 8, /* ncclUint8 */ 1, 0, comm, nullptr), "ncclBroadcast (length)");
    if (hipMemcpy(&len, d_len, 8, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy failed");      // (synchronises with the null stream)
    if (len == 0 || len > (1ull << 32)) die("--index-via rccl: implausible blob length");
    if (hipMalloc((void**)&d_blob, len) != hipSuccess) die("hipMalloc failed");
    if (rank == 0 && hipMemcpy(d_blob, blob_in.data(), len, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy failed");
    ok(bcast(d_blob, d_blob, len, 1, 0, comm, nullptr), "ncclBroadcast (blob)");
    std::string blob(len, '\0');
    if (hipMemcpy(&blob[0], d_blob, len, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy failed");
    (void)hipFree(d_blob); (void)hipFree(d_len); ok(destroy(comm), "ncclCommDestroy");
    fprintf(stderr, "stitch-align: rank %d of %d: reference index (%llu bytes) %s over RCCL\n", rank, world, len, rank == 0 ? "sent" : "received");
    return blob;
}

uint64_t fnv1a64(const std::string& s) { uint64_t h = 1469598103934665603ull; for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; } return h; }

}  // namespace

int main(int argc, char** argv) {
    signal(SIGPIPE, SIG_IGN);            // a closed output pipe is reported by the failing write, not by a silent death
    Args a = parse(argc, argv);
    Out out; out.bam = a.out_format == "bam"; out.level = a.compression;
    BamEncoder enc;

    if (!a.convert_sam.empty()) {                       // SAM text -> BAM (encoder self-test, no GPU)
        out.bam = true;
        LineReader in(a.convert_sam); std::string l, text; std::vector<std::string> recs;
        while (in.line(l)) {
            if (l.empty()) continue;
            if (l[0] == '@') { text += l + "\n"; if (l.rfind("@SQ", 0) == 0) { std::string sn; uint32_t ln = 0; size_t p = 0;
                    while (p < l.size()) { size_t t = l.find('\t', p); std::string f = l.substr(p, t == std::string::npos ? t : t - p); if (f.rfind("SN:", 0) == 0) sn = f.substr(3); if (f.rfind("LN:", 0) == 0) ln = (uint32_t)strtoul(f.c_str() + 3, nullptr, 10); if (t == std::string::npos) break; p = t + 1; }
                    enc.ref_id[sn] = (int)enc.refs.size(); enc.refs.push_back({sn, ln}); } }
            else recs.push_back(l);
        }
        enc.header(out, text);
        for (auto& r : recs) enc.record(out, r);
        out.finish();
        return 0;
    }

    if (a.ref_fasta.empty()) die(std::string("--ref-fasta is required\n") + USAGE);
    if (a.reads_fastq.empty() == a.reads_fasta.empty()) die("Must specify exactly one of --reads-fastq or --reads-fasta");

    // reference (target_seq::from_fasta): name = first word, sequence upper-cased by the library
    std::vector<std::string> names, seqs;
    { FastxReader fr(a.ref_fasta, false); Rec r; while (fr.next(r)) { const std::string nm = first_word(r.head); if (nm.empty()) die("empty read name"); names.push_back(nm); seqs.push_back(r.seq); } }
    if (names.empty()) die("Found no sequences in the FASTA");
    for (size_t k = 0; k < names.size(); ++k) { enc.ref_id[names[k]] = (int)k; enc.refs.push_back({names[k], (uint32_t)seqs[k].size()}); }
    const bool worker = a.shard_lo >= 0;
    const std::string header = sam_header(enc.refs, argc, argv);
    if (!worker) { if (out.bam) enc.header(out, header); else out.put(header.data(), header.size()); }

    const bool fastq = !a.reads_fastq.empty();
    FastxReader reads(fastq ? a.reads_fastq : a.reads_fasta, fastq);
    if (a.dry_run) { Rec r; uint64_t n = 0, bases = 0; while (reads.next(r)) { ++n; bases += r.seq.size(); } out.finish(); fprintf(stderr, "stitch-align: %zu targets, %llu reads, %llu bases\n", names.size(), (unsigned long long)n, (unsigned long long)bases); return 0; }

    std::vector<const char*> cn; std::vector<const uint8_t*> cs; std::vector<uint32_t> cl;
    for (size_t k = 0; k < names.size(); ++k) { cn.push_back(names[k].c_str()); cs.push_back((const uint8_t*)seqs[k].data()); cl.push_back((uint32_t)seqs[k].size()); }
    stitch_index* index = nullptr;
    if (!a.index_blob.empty() || a.rccl_rank >= 0) {    // worker: the index the parent built, as the serialized blob
        std::string blob;
        if (a.rccl_rank <= 0) {                         // (file hand-over, or rank 0 of the RCCL broadcast)
            FILE* f = fopen(a.index_blob.c_str(), "rb"); if (!f) die("cannot open " + a.index_blob);
            char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.append(buf, n); fclose(f);
        }
        if (a.rccl_rank >= 0) {
            if (hipSetDevice(a.device) != hipSuccess) die("bad device ordinal");
            // (RCCL announces its version on stdout, which carries the records: its chatter goes to stderr)
            fflush(stdout); const int saved = dup(1); dup2(2, 1);
            blob = rccl_broadcast_blob(a.rccl_rank, a.rccl_world, a.rccl_id, blob);
            fflush(stdout); dup2(saved, 1); close(saved);
        }
        if (stitch_index_deserialize(blob.data(), blob.size(), &index) != STITCH_OK) die(stitch_last_error());
    } else if (stitch_index_build(cn.data(), cs.data(), cl.data(), (uint32_t)names.size(), &index) != STITCH_OK) die(stitch_last_error());

    if (!a.devices.empty()) {
        // ---- parent of a multi-device run.  NOTHING in this block may call a device function: the workers are started with fork +
        // exec, and replacing a process that has initialised the GPU takes the machine down on some hosts (the library's host-only
        // entry points used here — index build / serialize, stitch_shard_range — never touch a device).
        const std::string path = fastq ? a.reads_fastq : a.reads_fasta;
        if (path == "-") die("--devices needs the reads in a file (every worker reads its own block)");
        // one pass over the reads, nothing kept but 20 bytes per record: where it starts, and (length, hash) to find the read groups
        std::string surrogate; std::vector<uint64_t> soffs(1, 0); std::vector<long long> rec_at;
        { Rec r; while (reads.next(r)) { const uint64_t hsh = fnv1a64(r.seq); const uint32_t len = (uint32_t)r.seq.size(); surrogate.append((const char*)&hsh, 8); surrogate.append((const char*)&len, 4);
                                          soffs.push_back(surrogate.size()); rec_at.push_back(reads.rec_off); } }
        const uint32_t n_all = (uint32_t)rec_at.size(), W = (uint32_t)a.devices.size();
        if (a.index_via == "rccl") { std::vector<int> d = a.devices; std::sort(d.begin(), d.end()); if (std::adjacent_find(d.begin(), d.end()) != d.end()) die("--index-via rccl needs one GPU per worker (RCCL refuses two ranks on one device)"); }
        struct Run {                                    // whatever happens, no worker and no temporary file is left behind
            std::string dir; std::vector<std::string> files; std::vector<pid_t> pids;
            void cleanup() { for (pid_t& p : pids) if (p > 0) { kill(p, SIGTERM); int st; waitpid(p, &st, 0); p = -1; } for (auto& f : files) unlink(f.c_str()); files.clear(); if (!dir.empty()) rmdir(dir.c_str()); dir.clear(); }
            ~Run() { cleanup(); }
        } run;
        char tmpl[] = "/tmp/stitch-align-XXXXXX"; const char* dir = mkdtemp(tmpl); if (!dir) die("mkdtemp failed");
        run.dir = dir;
        const std::string blob_path = run.dir + "/index.blob", id_path = run.dir + "/rccl.id";
        run.files = {blob_path, id_path, id_path + ".tmp"};
        { size_t len = 0; if (stitch_index_serialize(index, nullptr, &len) != STITCH_OK) die(stitch_last_error()); std::string blob(len, '\0');
          if (stitch_index_serialize(index, &blob[0], &len) != STITCH_OK) die(stitch_last_error());
          FILE* f = fopen(blob_path.c_str(), "wb"); if (!f || fwrite(blob.data(), 1, len, f) != len) die("cannot write " + blob_path); fclose(f); }
        std::vector<std::string> outs(W);
        for (uint32_t r = 0; r < W; ++r) { outs[r] = run.dir + "/rank" + std::to_string(r) + ".sam"; run.files.push_back(outs[r]); }
        bool failed = false; std::string why;
        for (uint32_t r = 0; r < W && !failed; ++r) {
            uint32_t lo = 0, hi = 0;
            if (stitch_shard_range((const uint8_t*)surrogate.data(), soffs.data(), n_all, W, r, &lo, &hi) != STITCH_OK) { failed = true; why = stitch_last_error(); break; }
            std::vector<std::string> av;
            // (the parent's own placement and output options are replaced; none of them has a short spelling, and `--flag=value` is refused by the parser)
            for (int i = 0; i < argc; ++i) { const std::string k = argv[i]; if (k == "--devices" || k == "--output-format" || k == "--device" || k == "--index-via") { ++i; continue; } av.push_back(k); }
            av.insert(av.end(), {"--device", std::to_string(a.devices[r]), "--shard", std::to_string(lo), std::to_string(hi), "--shard-offset", std::to_string(lo < n_all ? rec_at[lo] : 0),
                                 "--shard-out", outs[r], "--index-blob", blob_path});
            if (a.index_via == "rccl") av.insert(av.end(), {"--rccl", std::to_string(r), std::to_string(W), id_path});
            const pid_t pid = fork();
            if (pid < 0) { failed = true; why = "fork failed"; break; }
            if (pid == 0) { std::vector<char*> cv; for (auto& x : av) cv.push_back(&x[0]); cv.push_back(nullptr); execv("/proc/self/exe", cv.data()); _exit(127); }
            run.pids.push_back(pid);
        }
        // Reap the workers in whatever order they end: with --index-via rccl a worker that dies early (bad ordinal, hipMalloc) leaves the
        // others inside ncclCommInitRank / ncclBroadcast, whose bootstrap has no timeout — waiting for rank 0 first would hang for good.
        // The first failure ends the rest (cleanup() below: SIGTERM, then reaped).
        for (size_t left = run.pids.size(); left > 0 && !failed;) {
            int st = 0; const pid_t p = waitpid(-1, &st, 0);
            if (p < 0) { if (errno == EINTR) continue; failed = true; why = "waitpid failed"; break; }
            const auto it = std::find(run.pids.begin(), run.pids.end(), p);
            if (it == run.pids.end()) continue;          // (not one of the workers)
            const size_t rank = (size_t)(it - run.pids.begin());
            *it = -1; --left;
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { failed = true; why = "worker " + std::to_string(rank) + " failed (" + (WIFEXITED(st) ? "exit status " + std::to_string(WEXITSTATUS(st)) : "signal " + std::to_string(WTERMSIG(st))) + ")"; }
        }
        if (failed) { run.cleanup(); die(why); }         // (exit() does not unwind: clean up first)
        for (uint32_t r = 0; r < W; ++r) {                 // records in rank order = input order
            LineReader in(outs[r]); std::string l;
            while (in.line(l)) { if (l.empty()) continue; if (out.bam) enc.record(out, l); else { out.put(l.data(), l.size()); out.put("\n", 1); } }
        }
        out.finish();
        fprintf(stderr, "stitch-align: %u reads on %u devices\n", n_all, W);
        stitch_index_destroy(index);
        return 0;
    }
    if (worker) { if (!freopen(a.shard_out.c_str(), "w", stdout)) die("cannot write " + a.shard_out); out.bam = false; }
    // test hook: the worker of this device ordinal stalls before it touches a device, like a rank held inside an RCCL bootstrap whose
    // partner has died (tests/test_cli.py: the parent must end it and return promptly)
    if (worker) { const char* st = getenv("STITCH_ALIGN_TEST_STALL_DEVICE"); if (st && atoi(st) == a.device) for (;;) pause(); }
    stitch_ctx* ctx = nullptr;
    if (stitch_ctx_create(a.device, index, &a.o, &ctx) != STITCH_OK) die(stitch_last_error());

    // Three stages side by side, as the reference runs reader, aligners and writer concurrently (commands/align.rs:338-441): a reader thread
    // parses the next batches, this thread hands batch k to the device, a writer thread formats and writes batch k - 1.  The library's
    // result arrays live until the next stitch_align_batch, so the writer works on a copy of them through stitch_format_sam_chains (the
    // same formatter on caller-supplied chains); output order = input order (one writer, batches in sequence).
    struct Work { std::vector<Rec> recs; std::vector<stitch_read_result> rr; std::vector<stitch_chain> ch; std::vector<stitch_op> ops; };
    struct Chan {                                       // a bounded queue of batches between two stages
        std::mutex mu; std::condition_variable cv; std::deque<Work> q; bool closed = false; size_t cap = 2;
        void push(Work&& w) { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(w)); cv.notify_all(); }
        bool pop(Work& w) { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; w = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
        void close() { std::lock_guard<std::mutex> l(mu); closed = true; cv.notify_all(); }
    } to_align, to_write;
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    uint64_t n_reads = 0, cells_total = 0;
    double t_parse = 0, t_device = 0, t_format = 0, t_device_first = 0; uint64_t n_first = 0;      // (the first call also sizes and allocates the device arena: seconds)
    const auto t_run0 = clk::now();
    const bool serial = getenv("STITCH_ALIGN_SERIAL") != nullptr;      // (debugging) the three stages one after the other on this thread
    if (serial) { to_align.cap = ~(size_t)0; to_write.cap = ~(size_t)0; }
    auto reader_body = [&]() {
        Work w; Rec r; long rec_no = 0;
        if (worker && a.shard_offset > 0) { reads.seek(a.shard_offset); rec_no = a.shard_lo; }      // (no re-parsing of the blocks before this worker's)
        auto t0 = clk::now();
        while (reads.next(r)) {
            const long k = rec_no++;
            if (worker && (k < a.shard_lo || k >= a.shard_hi)) { if (k >= a.shard_hi) break; continue; }
            w.recs.push_back(r);
            if (w.recs.size() >= a.batch) { t_parse += secs(t0, clk::now()); to_align.push(std::move(w)); w = Work(); t0 = clk::now(); }
        }
        t_parse += secs(t0, clk::now());
        if (!w.recs.empty()) to_align.push(std::move(w));
        to_align.close();
    };
    std::thread reader; if (serial) reader_body(); else reader = std::thread(reader_body);
    std::vector<const char*> tn; std::vector<uint32_t> tl;
    for (size_t k = 0; k < names.size(); ++k) { tn.push_back(names[k].c_str()); tl.push_back((uint32_t)seqs[k].size()); }
    auto writer_body = [&]() {
        Work w; std::vector<char> text(1 << 20);
        while (to_write.pop(w)) {
            const auto t0 = clk::now();
            for (size_t k = 0; k < w.recs.size(); ++k) {
                const Rec& r = w.recs[k]; const stitch_read_result& R = w.rr[k];
                long len;
                for (;;) {
                    len = stitch_format_sam_chains(&a.o, tn.data(), tl.data(), (uint32_t)tn.size(), r.head.c_str(), (const uint8_t*)r.seq.data(),
                                                   r.has_qual ? (const uint8_t*)r.qual.data() : nullptr, r.seq.size(), w.ch.data() + R.chains_begin, R.n_chains, w.ops.data(),
                                                   R.has_prealign, R.prealign_score, text.data(), text.size());
                    if (len < 0) die(stitch_last_error());
                    if ((size_t)len < text.size()) break;
                    text.resize((size_t)len + 1);
                }
                if (!out.bam) { out.put(text.data(), (size_t)len); if (len) out.put("\n", 1); }
                else { size_t p = 0; const std::string all(text.data(), (size_t)len); while (p < all.size()) { size_t e = all.find('\n', p); if (e == std::string::npos) e = all.size(); if (e > p) enc.record(out, all.substr(p, e - p)); p = e + 1; } }
            }
            if (!out.bam) fflush(stdout);
            t_format += secs(t0, clk::now());
        }
    };
    std::thread writer; if (!serial) writer = std::thread(writer_body);
    {
        Work w;
        while (to_align.pop(w)) {
            const auto t0 = clk::now();
            std::string cat; std::vector<uint64_t> offs(w.recs.size() + 1, 0);
            for (size_t k = 0; k < w.recs.size(); ++k) { cat += w.recs[k].seq; offs[k + 1] = cat.size(); }
            const stitch_read_result* rr; const stitch_chain* ch; const stitch_op* ops; uint64_t cells = 0;
            if (stitch_align_batch(ctx, (const uint8_t*)cat.data(), offs.data(), (uint32_t)w.recs.size(), &rr, &ch, &ops, &cells) != STITCH_OK) die(stitch_last_error());
            cells_total += cells; n_reads += w.recs.size();
            // (the copy the writer formats from: the arrays themselves are the library's until the next call)
            w.rr.assign(rr, rr + w.recs.size());
            size_t n_ch = 0, n_ops = 0;
            for (const stitch_read_result& R : w.rr) n_ch = std::max<size_t>(n_ch, (size_t)R.chains_begin + R.n_chains);
            w.ch.assign(ch, ch + n_ch);
            for (const stitch_chain& c2 : w.ch) n_ops = std::max<size_t>(n_ops, (size_t)c2.ops_begin + c2.ops_len);
            w.ops.assign(ops, ops + n_ops);
            const double dt_call = secs(t0, clk::now());
            if (t_device == 0) { t_device_first = dt_call; n_first = w.recs.size(); }
            t_device += dt_call;
            to_write.push(std::move(w)); w = Work();
        }
        to_write.close();
    }
    if (serial) writer_body(); else { reader.join(); writer.join(); }
    const double t_all = secs(t_run0, clk::now());
    out.finish();
    fprintf(stderr, "stitch-align: %llu reads, %.3f Gcells in %.2f s = %.1f reads/s end to end (reader %.2f s, device calls %.2f s, formatter + writer %.2f s, side by side; first call %.2f s for %llu reads, the others %.1f reads/s)\n",
            (unsigned long long)n_reads, cells_total / 1e9, t_all, n_reads / std::max(t_all, 1e-9), t_parse, t_device, t_format, t_device_first, (unsigned long long)n_first,
            n_reads > n_first ? (n_reads - n_first) / std::max(t_device - t_device_first, 1e-9) : 0.0);
    stitch_ctx_destroy(ctx); stitch_index_destroy(index);
    return 0;
}
