// ngm_cli.cpp -- `ngm-hip`: NextGenMap-compatible command line over the C ABI (include/ngm_pipeline.h).
//
// Mirrors what ngm-core does around its hot path:
//   option names / defaults          src/config/Options.h:13-105, src/config/Config.cpp:381-557
//   read parsing                     src/parser/IParser.h:59-121 (upper-case, non-ACGT -> N, truncate to qry_max_len-1)
//   parameter estimation             src/ReadProvider.cpp:163-394 (qry_max_len, corridor, sensitivity)
//   output filter                    src/writer/GenericReadWriter.h:190-254 (min_identity, min_residues), min_mq
//   SAM header / records             src/writer/SAMWriter.cpp:17-84, :98-228, :312-373
//   batch hand-out to workers        src/NGM.cpp:232-279 (GetNextReadBatch under a mutex), src/CS.cpp:440-456 (-g: threads -> devices)
//   buffered, ordered output         src/writer/GenericReadWriter.h:190-304 (20 MB buffer, flushed under the output lock)
// Everything heavy (index, candidate search, score, align) happens on the GPU behind ngm_mapper_*.
// Single-end and paired-end (-p -q interleaved, --qry1/--qry2), --affine, -n/--strata, SAM and BAM (--bam) output, one or
// several GPUs (-g 0,1,...).  Not supported (rejected loudly): bisulfite / SLAM-seq, --argos, --vcf, SAM/BAM *input*.
//
// Pass 2 is a pipeline, not a loop:
//   splitter (1 thread)   cuts the input into batches: for plain 4-line FASTQ it only counts line ends in the mapped file
//                         (record boundaries every 1 024 reads), for gz / FASTA / multi-line input it is the serial kseq-style
//                         reader (inflate is one stream; that is the bound of such input, as in the reference);
//   workers               `--workers` per GPU (default 2), each owning an ngm_mapper like a NextGenMap CS thread owns its
//                         IAlignment: parse its batch into a page-locked row buffer (pool threads, zero-copy names / qualities),
//                         map it on its GPU, format the SAM text (pool threads) -- while one worker's batch is on the GPU the
//                         others parse and format;
//   writer (1 thread)     batches in input order, large writes.
// Paired-end tie-breaks see the same running mean insert size as `ngm -t 1` (ngm_pair_state: batches take turns for that
// part only), so the output does not depend on the number of workers or GPUs.
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <ucontext.h>
#include <malloc.h>
#include <getopt.h>
#include <sys/mman.h>
#include <poll.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/sendfile.h>
#include <sys/wait.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <unistd.h>
#include <vector>

#include "../../include/ngm_hip.h"
#include "../../include/ngm_pipeline.h"
#include "bam_writer.h"
#include "gz_inflate.h"
#include "thread_pool.h"

namespace {

struct Read { std::string name, seq, qual; uint8_t unit = 0; };   // unit: --broken-pairs (1 a read without its mate, 2 the placeholder in the mate's row, 4 the pair's 0x40 / 0x80 flags are swapped)

// FASTA / FASTQ reader with kseq semantics: name = header up to the first white space, multi-line records
class SeqReader {
public:
	explicit SeqReader(const char *path) : f_(gzopen(path, "rb")) { if (f_) gzbuffer(f_, 1 << 20); }
	~SeqReader() { if (f_) gzclose(f_); }
	bool ok() const { return f_ != nullptr; }
	bool next(Read &r) {
		std::string line;
		if (!have_header_) { while (getline(line)) if (!line.empty() && (line[0] == '@' || line[0] == '>')) { header_ = line; have_header_ = true; break; } }
		if (!have_header_) return false;
		const bool fastq = header_[0] == '@';
		size_t e = 1;
		while (e < header_.size() && header_[e] != ' ' && header_[e] != '\t') ++e;
		r.name.assign(header_, 1, e - 1);
		r.seq.clear();
		r.qual.clear();
		have_header_ = false;
		while (getline(line)) {
			if (fastq && !line.empty() && line[0] == '+') break;
			if (!fastq && !line.empty() && (line[0] == '>' || line[0] == '@')) { header_ = line; have_header_ = true; break; }
			r.seq += line;
		}
		if (fastq) {
			while (r.qual.size() < r.seq.size() && getline(line)) r.qual += line;
			if (r.qual.size() != r.seq.size()) { fprintf(stderr, "[ngm-hip] error: Error while parsing read: sequence and quality lengths differ (%s)\n", r.name.c_str()); exit(1); }
		}
		return true;
	}

private:
	bool getline(std::string &out) {
		out.clear();
		char buf[65536];
		bool any = false;
		while (gzgets(f_, buf, sizeof(buf))) {
			any = true;
			size_t n = strlen(buf);
			const bool eol = n && buf[n - 1] == '\n';
			while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
			out.append(buf, n);
			if (eol) break;
		}
		return any;
	}
	gzFile f_;
	std::string header_;
	bool have_header_ = false;
};

struct Opts {
	std::string ref, qry, qry1, qry2, out;
	int paired = 0, min_insert = 0, max_insert = 1000, topn = 1, strata = 0;
	char pe_delimiter = '/';
	int device = 0, kmer = 13, kmer_skip = 2, bin_size = 2, mode = 0, corridor = -1, max_read_length = 0, min_mq = 0, max_kfreq = 0;
	int match = 10, mismatch = 15, gap_read = -1, gap_ref = -1, gap_extend = -1, affine = 0, hard_clip = 0, silent_clip = 0, no_unal = 0, fast_pairing = 0, broken_pairs = 0, max_cmrs = 2147483647;
	int skip_save = 0, bam = 0, workers = 2, serial_reader = 0;
	int bs_mapping = 0, bs_cutoff = 6, match_tt = -1, match_tc = -1, match_set = 0, mismatch_set = 0, slam_seq = 0;
	std::vector<int> devices;
	int shard_i = 0, shard_n = 1, shard_output = 0, keep_shards = 0;
	std::string rg[12];  // read group: ID CN DS DT FO KS LB PG PI PL PU SM (SAMWriter.cpp:46-80)
	int very_fast = 0, fast = 0, sensitive = 0, very_sensitive = 0, variant = NGM_VARIANT_OCL_GPU;
	float sensitivity = -1.f, kmer_min = 0.f, min_identity = 0.65f, min_residues = 0.5f;
	int batch = 1 << 18;
	int stats_fd = -1;   // --stats-fd (set by the parent of shard processes): this shard's int64[8] statistics go there
	int ref_score_buffer = -1;   // --reference-score-buffer: entries of NextGenMap's score buffer whose lost pairs are mirrored (-1: 1 024 with --affine, else 0 = none)
	std::string cmdline;
};

[[noreturn]] void die(const std::string &msg) { fprintf(stderr, "[ngm-hip] error: %s\n", msg.c_str()); exit(1); }
// (one fprintf per message: stdio locks the stream for the call, so the lines of the reader, the writer and the estimate's thread never interleave)
void info(const char *tag, const std::string &msg) { fprintf(stderr, "[%s] %s\n", tag, msg.c_str()); }

Opts parse(int argc, char **argv) {
	Opts o;
	for (int i = 1; i < argc; ++i) { if (i > 1) o.cmdline += " "; o.cmdline += argv[i]; }  // Config.cpp:565-574
	enum { KSKIP = 1000, HARD, SILENT, KMIN, MB, MMP, GRP, GFP, MAXCMRS, NOUNAL, NOPROG, MAXRL, BINSZ, MAXKF, VFAST, FAST, SENS, VSENS, DEVICE,
		SKIPSAVE, BATCH, VARIANT, SHARD, SHARDOUT, KEEPSHARDS, BAMOUT, WORKERS, SERIAL, AFFINE, GEP, PEDELIM, STRATA, BSMAP, BSCUT, MBTT, MBTC, SLAM, FASTPAIR, BROKENPAIRS, REFSCOREBUF, STATSFD, RG0, RG_LAST = RG0 + 11, UNSUPPORTED };
	static const option lo[] = {
		{"ref", required_argument, 0, 'r'}, {"qry", required_argument, 0, 'q'}, {"output", required_argument, 0, 'o'},
		{"cpu-threads", required_argument, 0, 't'}, {"gpu", no_argument, 0, 'g'}, {"sensitivity", required_argument, 0, 's'},
		{"kmer", required_argument, 0, 'k'}, {"kmer-skip", required_argument, 0, KSKIP}, {"max-consec-indels", required_argument, 0, 'C'},
		{"local", no_argument, 0, 'l'}, {"end-to-end", no_argument, 0, 'e'}, {"min-identity", required_argument, 0, 'i'},
		{"min-residues", required_argument, 0, 'R'}, {"min-mq", required_argument, 0, 'Q'}, {"hard-clip", no_argument, 0, HARD},
		{"silent-clip", no_argument, 0, SILENT}, {"kmer-min", required_argument, 0, KMIN}, {"match-bonus", required_argument, 0, MB},
		{"mismatch-penalty", required_argument, 0, MMP}, {"gap-read-penalty", required_argument, 0, GRP}, {"gap-ref-penalty", required_argument, 0, GFP},
		{"max-cmrs", required_argument, 0, MAXCMRS}, {"no-unal", no_argument, 0, NOUNAL}, {"no-progress", no_argument, 0, NOPROG},
		{"max-read-length", required_argument, 0, MAXRL}, {"bin-size", required_argument, 0, BINSZ}, {"max-kfreq", required_argument, 0, MAXKF},
		{"very-fast", no_argument, 0, VFAST}, {"fast", no_argument, 0, FAST}, {"sensitive", no_argument, 0, SENS}, {"very-sensitive", no_argument, 0, VSENS},
		{"device", required_argument, 0, DEVICE}, {"skip-save", no_argument, 0, SKIPSAVE}, {"batch-size", required_argument, 0, BATCH},
		{"kernel-variant", required_argument, 0, VARIANT}, {"shard", required_argument, 0, SHARD}, {"shard-output", no_argument, 0, SHARDOUT},
		{"keep-shards", no_argument, 0, KEEPSHARDS},
		{"qry1", required_argument, 0, '1'}, {"qry2", required_argument, 0, '2'}, {"paired", no_argument, 0, 'p'},
		{"min-insert-size", required_argument, 0, 'I'}, {"max-insert-size", required_argument, 0, 'X'}, {"pe-delimiter", required_argument, 0, PEDELIM},
		{"rg-id", required_argument, 0, RG0}, {"rg-cn", required_argument, 0, RG0 + 1}, {"rg-ds", required_argument, 0, RG0 + 2},
		{"rg-dt", required_argument, 0, RG0 + 3}, {"rg-fo", required_argument, 0, RG0 + 4}, {"rg-ks", required_argument, 0, RG0 + 5},
		{"rg-lb", required_argument, 0, RG0 + 6}, {"rg-pg", required_argument, 0, RG0 + 7}, {"rg-pi", required_argument, 0, RG0 + 8},
		{"rg-pl", required_argument, 0, RG0 + 9}, {"rg-pu", required_argument, 0, RG0 + 10}, {"rg-sm", required_argument, 0, RG0 + 11},
		{"fast-pairing", no_argument, 0, FASTPAIR}, {"broken-pairs", no_argument, 0, BROKENPAIRS}, {"reference-score-buffer", required_argument, 0, REFSCOREBUF}, {"stats-fd", required_argument, 0, STATSFD},
		{"affine", no_argument, 0, AFFINE}, {"gap-extend-penalty", required_argument, 0, GEP}, {"bam", no_argument, 0, BAMOUT}, {"workers", required_argument, 0, WORKERS}, {"serial-reader", no_argument, 0, SERIAL}, {"bs-mapping", no_argument, 0, BSMAP},
		{"bs-cutoff", required_argument, 0, BSCUT}, {"match-bonus-tt", required_argument, 0, MBTT}, {"match-bonus-tc", required_argument, 0, MBTC},
		{"slam-seq", required_argument, 0, SLAM}, {"topn", required_argument, 0, 'n'}, {"strata", no_argument, 0, STRATA},
		{"argos", no_argument, 0, UNSUPPORTED}, {"vcf", required_argument, 0, UNSUPPORTED}, {"config", required_argument, 0, UNSUPPORTED},
		{0, 0, 0, 0}};
	int c, idx = 0;
	while ((c = getopt_long(argc, argv, "o:q:r:t:gs:k:lei:R:C:Q:p1:2:I:X:n:", lo, &idx)) != -1) {
		switch (c) {
		case 'r': o.ref = optarg; break;
		case 'q': o.qry = optarg; break;
		case '1': o.qry1 = optarg; break;
		case '2': o.qry2 = optarg; break;
		case 'p': o.paired = 1; break;
		case 'n': o.topn = std::max(1, atoi(optarg)); break;
		case STRATA: o.strata = 1; break;
		case 'I': o.min_insert = atoi(optarg); break;
		case 'X': o.max_insert = atoi(optarg); break;
		case PEDELIM: o.pe_delimiter = optarg[0]; break;
		case 'o': o.out = optarg; break;
		case 't': break;  // host threads follow the machine (NGM_HIP_HOST_THREADS)
		case 'g':  // "-g" or "-g 0,1,..." (optional list without '=', src/config/Config.cpp:624-647); the GPU is not optional here
			if (optind < argc && argv[optind][0] >= '0' && argv[optind][0] <= '9') {
				o.devices.clear();
				for (const char *p = argv[optind]; *p;) { o.devices.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
				++optind;
			}
			break;
		case 's': o.sensitivity = (float) atof(optarg); break;
		case 'k': o.kmer = atoi(optarg); break;
		case KSKIP: o.kmer_skip = atoi(optarg); break;
		case 'C': o.corridor = 2 * atoi(optarg); break;  // Config.cpp:540-557
		case 'l': o.mode = 0; break;
		case 'e': o.mode = 1; break;
		case 'i': o.min_identity = (float) atof(optarg); break;
		case 'R': o.min_residues = (float) atof(optarg); break;
		case 'Q': o.min_mq = atoi(optarg); break;
		case HARD: o.hard_clip = 1; break;
		case SILENT: o.silent_clip = 1; break;
		case KMIN: o.kmer_min = (float) atof(optarg); break;
		case MB: o.match = atoi(optarg); o.match_set = 1; break;
		case MMP: o.mismatch = atoi(optarg); o.mismatch_set = 1; break;
		case BSMAP: o.bs_mapping = 1; break;
		case SLAM: o.slam_seq = atoi(optarg); break;
		case BSCUT: o.bs_cutoff = atoi(optarg); break;
		case MBTT: o.match_tt = atoi(optarg); break;
		case MBTC: o.match_tc = atoi(optarg); break;
		case GRP: o.gap_read = atoi(optarg); break;
		case GFP: o.gap_ref = atoi(optarg); break;
		case AFFINE: o.affine = 1; break;
		case RG0: case RG0 + 1: case RG0 + 2: case RG0 + 3: case RG0 + 4: case RG0 + 5: case RG0 + 6: case RG0 + 7: case RG0 + 8: case RG0 + 9:
		case RG0 + 10: case RG0 + 11: o.rg[c - RG0] = optarg; break;
		case GEP: o.gap_extend = atoi(optarg); break;
		case MAXCMRS: o.max_cmrs = atoi(optarg); break;
		case NOUNAL: o.no_unal = 1; break;
		case NOPROG: break;
		case SKIPSAVE: o.skip_save = 1; break;
		case MAXRL: o.max_read_length = atoi(optarg); break;
		case BINSZ: o.bin_size = atoi(optarg); break;
		case MAXKF: o.max_kfreq = atoi(optarg); break;
		case VFAST: o.very_fast = 1; break;
		case FAST: o.fast = 1; break;
		case SENS: o.sensitive = 1; break;
		case VSENS: o.very_sensitive = 1; break;
		case DEVICE: o.device = atoi(optarg); o.devices.assign(1, o.device); break;
		case BAMOUT: o.bam = 1; break;
		case WORKERS: o.workers = std::max(1, atoi(optarg)); break;
		case SERIAL: o.serial_reader = 1; break;
		case BATCH: o.batch = std::max(1024, atoi(optarg)); break;
		case VARIANT: o.variant = atoi(optarg) ? NGM_VARIANT_OCL_CPU : NGM_VARIANT_OCL_GPU; break;
		case SHARD: {  // "i/N": this process maps the i-th of N contiguous ranges of the input and writes their records (the header with shard 0)
			const char *sl = strchr(optarg, '/');
			o.shard_i = atoi(optarg); o.shard_n = sl ? atoi(sl + 1) : 0;
			if (!sl || o.shard_n < 1 || o.shard_i < 0 || o.shard_i >= o.shard_n) die("--shard expects i/N with 0 <= i < N");
			break;
		}
		case SHARDOUT: o.shard_output = 1; break;
		case KEEPSHARDS: o.keep_shards = 1; break;
		case FASTPAIR: o.fast_pairing = 1; break;
		case REFSCOREBUF: o.ref_score_buffer = std::max(0, atoi(optarg)); break;
		case STATSFD: o.stats_fd = atoi(optarg); break;
		case BROKENPAIRS: o.broken_pairs = 1; break;
		case UNSUPPORTED: die(std::string("option --") + lo[idx].name + " is not supported by the HIP backend yet");
		default: die("unknown option (see src/config/Options.h of NextGenMap for the option set)");
		}
	}
	if (o.ref.empty()) die("no reference given (-r/--ref)");
	if (!o.qry1.empty() && !o.qry2.empty()) o.paired = 1;  // Config.cpp:395-399
	else if (!o.qry1.empty() || !o.qry2.empty()) die("--qry1 and --qry2 must be given together");
	if (o.paired && o.topn > 1) die("Paired end mode with topn > 1 not yet supported.");  // ScoreBuffer::topNPE
	if (o.paired && o.qry.empty() && o.qry1.empty()) die("-p/--paired needs -q (interleaved mates) or --qry1/--qry2");
	// scoring defaults depend on the personality (Config.cpp:433-470)
	if (o.bs_mapping) {
		info("MAIN", "Using bs-mapping scoring scheme");
		if (o.affine) die("'--bs-mapping' and '--affine' can't be used at the same time!");
		if (o.mode == 1) die("'--bs-mapping' and '--e/--end-to-end' can't be used at the same time!");
		if (!o.match_set) o.match = 4;
		if (!o.mismatch_set) o.mismatch = 2;
		if (o.gap_read < 0) o.gap_read = 10;
		if (o.gap_ref < 0) o.gap_ref = 10;
		if (o.gap_extend < 0) o.gap_extend = 2;
		if (o.match_tt < 0) o.match_tt = 4;
		if (o.match_tc < 0) o.match_tc = 4;
	}
	if (o.slam_seq) {
		if (o.bs_mapping) die("'--bs-mapping' and '--slam-seq' can't be used at the same time!");   // Config.cpp:454-457
		if (o.affine) die("'--slam-seq' needs the default (linear-gap) scoring: the affine backend produces no per-base records");
	}
	if (o.match_tt < 0) o.match_tt = 10;
	if (o.match_tc < 0) o.match_tc = 2;
	if (o.gap_read < 0) o.gap_read = o.affine ? 33 : 20;
	if (o.gap_ref < 0) o.gap_ref = o.affine ? 33 : 20;
	if (o.gap_extend < 0) o.gap_extend = o.affine ? 3 : 5;
	// the pairs NextGenMap loses (ngm_mapper_set_reference_score_buffer): its SeqAn personality scores in batches of 1 024
	// (src/seqan/EndToEndAffine.h:44-46); the OpenCL personality's batch depends on the device it finds -- nothing to mirror by default
	// The walk that finds those pairs is sequential over the WHOLE input (it follows the reference's CS batches and score-buffer fill from
	// the first read on, as `ngm -t 1` does): a shard that starts in the middle of the input cannot know where the reference's buffer
	// stands at its first read, would lose pairs the reference keeps, and `cat shards` would no longer equal the single run.
	// Sharded runs therefore never mirror them.
	if (o.shard_n > 1) {
		if (o.ref_score_buffer > 0) fprintf(stderr, "[MAIN] --reference-score-buffer is ignored with --shard: the reference's score buffer cannot be followed from the middle of the input\n");
		o.ref_score_buffer = 0;
	}
	if (o.ref_score_buffer < 0) o.ref_score_buffer = o.affine ? 1024 : 0;
	if (o.devices.empty()) o.devices.assign(1, o.device);
	o.device = o.devices[0];
	if (o.broken_pairs) {
		if (!(o.paired && !o.qry.empty())) die("--broken-pairs only works with interleaved paired-end files.");   // ReadProvider.cpp:176-179
		if (o.shard_n > 1 || o.shard_output) die("--broken-pairs cannot be combined with --shard / --shard-output: the pairing of the records is decided while they are read");
		o.serial_reader = 1;   // which records are mates is decided record by record (ReadProvider.cpp:540-575)
	}
	return o;
}

// ---- input: records as views into the mapped file (plain FASTQ) or into storage owned by the batch (serial reader) ----
struct Rec { const char *name; const char *seq; const char *qual; uint32_t name_len, seq_len, qual_len; uint8_t unit; };

inline void pack_row_view(const char *seq, size_t len, int q, char *row) {  // IParser.h:59-121
	memset(row, 0, q);
	if (len == 0) { row[0] = 'N'; return; }
	const int L = (int) std::min<size_t>(len, (size_t) q - 1);
	for (int i = 0; i < L; ++i) {
		const char c = (char) (seq[i] & 0xDF);  // toupper for letters
		row[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
	}
}
void pack_row(const Read &r, int q, char *row) { pack_row_view(r.seq.data(), r.seq.size(), q, row); }

// a whole file mapped read-only; plain() = not gzip and made of 4-line FASTQ records (checked on the first records)
struct MappedFile {
	const char *p = nullptr;
	size_t n = 0, map_len = 0;   // n: the bytes that hold records (white space at the end of the file is not a record: kseq skips it too)
	int fd = -1;
	char *owned = nullptr;        // gzip input: the inflated text lives here instead of in a file mapping
	size_t owned_map = 0;         // != 0: `owned` is an anonymous mapping of this many bytes (gz_inflate.h), else malloc'ed
	bool open(const char *path) {
		fd = ::open(path, O_RDONLY);
		if (fd < 0) return false;
		struct stat st;
		if (fstat(fd, &st) != 0 || st.st_size == 0) return false;
		unsigned char magic[2] = {0, 0};
		if (st.st_size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return inflate_all(path);
		n = map_len = (size_t) st.st_size;
		void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m == MAP_FAILED) { p = nullptr; return false; }
		p = (const char *) m;
		madvise(m, map_len, MADV_SEQUENTIAL);
		trim();
		return n > 0;
	}
	void trim() { while (n > 0 && (p[n - 1] == '\n' || p[n - 1] == '\r' || p[n - 1] == ' ' || p[n - 1] == '\t')) --n; }
	// .gz input: inflated ONCE into memory (zlib, one thread per file -- the two files of a pair at the same time) and then read
	// like a mapped plain file by all pool threads, instead of twice through a line-by-line reader (estimation pass, mapping pass).
	// Inputs that inflate to more than 64 GB per file go through the serial reader.
	bool inflate_all(const char *path) {
		const size_t cap_max = (size_t) 64 << 30;
		{   // (what gz_inflate.h refuses -- a damaged stream, a wrong CRC or length -- falls back to zlib's inflate below)
			char *text = nullptr;
			size_t len = 0, reserved = 0;
			if (ngm::gz::inflate_file(path, &text, &len, &reserved, cap_max)) {
				owned = text; owned_map = reserved; p = text; n = len; map_len = 0;
				trim();
				return n > 0;
			}
		}
		gzFile g = gzopen(path, "rb");
		if (!g) return false;
		gzbuffer(g, 1 << 20);
		size_t cap = (size_t) 256 << 20, len = 0;
		char *buf = (char *) malloc(cap);
		bool ok = buf != nullptr;
		while (ok) {
			if (cap - len < ((size_t) 16 << 20)) {
				if (cap >= cap_max) { ok = false; break; }
				cap = std::min(cap_max, cap * 2);
				char *nb = (char *) realloc(buf, cap);
				if (!nb) { ok = false; break; }
				buf = nb;
			}
			const int got = gzread(g, buf + len, (unsigned) std::min<size_t>(cap - len, (size_t) 1 << 30));
			if (got < 0) { ok = false; break; }
			if (got == 0) break;
			len += (size_t) got;
		}
		gzclose(g);
		if (!ok || len == 0) { free(buf); return false; }
		owned = buf; p = buf; n = len; map_len = 0;
		trim();
		return n > 0;
	}
	void drop_owned() { if (owned_map) munmap(owned, owned_map); else free(owned); owned = nullptr; owned_map = 0; }
	void release() { if (owned) { drop_owned(); p = nullptr; n = 0; } }
	~MappedFile() { if (owned) drop_owned(); else if (p) munmap((void *) p, map_len); if (fd >= 0) close(fd); }
	// touch every page from the pool threads: the page-table entries of a multi-GB input are then set up in parallel instead
	// of one minor fault at a time under the (single) splitter thread
	void prefault() const {
		if (!p) return;
		const size_t step = (size_t) 8 << 20;
		const int chunks = (int) ((n + step - 1) / step);
		std::atomic<unsigned> sink{0};
		ngm::ThreadPool::instance().parallel_for(chunks, [&](int lo, int hi) {
			unsigned acc = 0;
			for (int c = lo; c < hi; ++c) for (size_t o = (size_t) c * step, e = std::min(n, o + step); o < e; o += 4096) acc += (unsigned char) p[o];
			sink += acc;
		}, 1);
	}
	// one 4-line record starting at `at`: sets the views, returns the offset of the next record, or 0 if malformed
	size_t record(size_t at, Rec &r) const {
		const char *b = p + at, *end = p + n;
		if (at >= n || *b != '@') return 0;
		const char *e0 = (const char *) memchr(b, '\n', end - b);
		if (!e0) return 0;
		const char *s = e0 + 1;
		const char *e1 = s < end ? (const char *) memchr(s, '\n', end - s) : nullptr;
		if (!e1) return 0;
		const char *pl = e1 + 1;
		if (pl >= end || *pl != '+') return 0;
		const char *e2 = (const char *) memchr(pl, '\n', end - pl);
		if (!e2) return 0;
		const char *ql = e2 + 1;
		const char *e3 = ql <= end ? (const char *) memchr(ql, '\n', end - ql) : nullptr;
		const char *qe = e3 ? e3 : end;  // the last line may lack its line end
		auto trim = [](const char *a, const char *z) { while (z > a && z[-1] == '\r') --z; return (uint32_t) (z - a); };
		const char *nm = b + 1;
		uint32_t nl = 0;
		const uint32_t hl = trim(nm, e0);
		while (nl < hl && nm[nl] != ' ' && nm[nl] != '\t') ++nl;  // kseq: name = header up to the first white space
		r.name = nm; r.name_len = nl; r.seq = s; r.seq_len = trim(s, e1); r.qual = ql; r.qual_len = trim(ql, qe);
		return (size_t) ((e3 ? e3 + 1 : end) - p);
	}
	bool plain_fastq() const {
		if (n < 2 || ((unsigned char) p[0] == 0x1f && (unsigned char) p[1] == 0x8b)) return false;
		size_t at = 0;
		Rec r;
		for (int i = 0; i < 1000 && at < n; ++i) {
			const size_t nx = record(at, r);
			if (!nx || r.qual_len != r.seq_len) return false;
			at = nx;
		}
		return true;
	}
};

// ---- a batch on its way through the pipeline -------------------------------------------------------------------------
struct Batch {
	uint64_t seq = 0;
	int n = 0;
	// plain input: byte offsets of every kSub-th record in file 0 / file 1 (two-file paired input: records alternate)
	std::vector<size_t> sub0, sub1;
	int n0 = 0, n1 = 0;                 // records from file 0 / file 1
	std::vector<Read> owned;            // serial reader: the records themselves
	std::vector<Rec> recs;
	std::vector<std::string> chunks;    // formatted output, in order
	char *text = nullptr;               // ... or, formatted on the GPU: one piece in a page-locked buffer of the pool
	size_t text_len = 0, text_cap = 0;
	size_t n_total = 0, n_mapped = 0, n_written = 0;
};
constexpr int kSub = 1024;

// ---- record index of a plain FASTQ file, built by all pool threads ----------------------------------------------------
// The file is cut into byte ranges; every range finds its first record (a line that starts with '@' whose second-next line
// starts with '+': in 4-line FASTQ only header lines have that -- a quality line that starts with '@' is followed by a header and a
// sequence), counts its records, and after a prefix sum over the ranges walks them again to note the byte offset of every
// `step`-th record of the FILE and, for the first input file, what ReadProvider's estimation pass collects (read lengths of the
// first 10 000 001 non-empty reads, every 1 000-th of the first 10 000 000 for the sensitivity sample; src/ReadProvider.cpp:201-305).
struct FastqIndex {
	bool ok = false;
	size_t bad_at = 0;                 // !ok: offset of the record that is not a 4-line FASTQ record
	std::string length_error;          // the first read whose sequence and quality lengths differ (IParser.h copyToRead), if any
	int step = kSub;
	std::vector<size_t> sub;           // offset of record 0, step, 2 step, ...; one more entry: the end of the last record
	size_t n_records = 0, n_nonempty = 0;
	size_t max_len = 0, min_len = 9999999, sum_len = 0, count = 0;   // the estimation pass's numbers
	std::vector<Read> sample;
};

void build_fastq_index(const MappedFile &f, int step, bool stats, FastqIndex &ix) {
	ngm::ThreadPool &pool = ngm::ThreadPool::instance();
	const int T = (int) std::max<size_t>(1, std::min<size_t>((size_t) pool.size() * 4, f.n >> 20));
	struct Range { size_t start = 0, n_rec = 0, n_ne = 0, bad_at = 0, rec_base = 0, ne_base = 0, max_len = 0, min_len = 9999999, sum_len = 0; bool bad = false; std::string len_err; std::vector<Read> sample; };
	std::vector<Range> rg((size_t) T + 1);
	rg[T].start = f.n;
	auto line_start_after = [&](size_t o) -> size_t {  // first line start at or after o
		if (o == 0 || f.p[o - 1] == '\n') return o;
		const char *c = (const char *) memchr(f.p + o, '\n', f.n - o);
		return c ? (size_t) (c + 1 - f.p) : f.n;
	};
	auto next_line = [&](size_t o) -> size_t { const char *c = o < f.n ? (const char *) memchr(f.p + o, '\n', f.n - o) : nullptr; return c ? (size_t) (c + 1 - f.p) : f.n; };
	pool.parallel_for(T, [&](int lo, int hi) {
		for (int r = lo; r < hi; ++r) {
			size_t at = r == 0 ? 0 : line_start_after(f.n / T * r);
			if (r > 0) {
				// a header line: '@' here and '+' two lines on
				for (; at < f.n; at = next_line(at)) {
					if (f.p[at] != '@') continue;
					const size_t l2 = next_line(next_line(at));
					if (l2 < f.n && f.p[l2] == '+') break;
				}
			}
			rg[r].start = at;
		}
	}, 1);
	for (int r = 1; r <= T; ++r) if (rg[r].start < rg[r - 1].start) rg[r].start = rg[r - 1].start;  // (ranges shorter than a record)
	auto walk = [&](int r, bool second) {
		Range &R = rg[r];
		Rec rec;
		size_t at = R.start, g = R.rec_base, c = R.ne_base;
		const size_t end = rg[r + 1].start;
		while (at < end) {
			const size_t nx = f.record(at, rec);
			if (!nx) { R.bad = true; R.bad_at = at; return; }
			if (!second) {
				if (rec.qual_len != rec.seq_len && R.len_err.empty()) R.len_err.assign(rec.name, rec.name_len);
				++R.n_rec;
				if (rec.seq_len) ++R.n_ne;
			} else {
				if (g % (size_t) step == 0) ix.sub[g / (size_t) step] = at;
				++g;
				if (stats && rec.seq_len) {   // reads without a sequence are not counted (ReadProvider.cpp:236)
					++c;
					if (c <= 10000001) {
						const size_t len = std::min<size_t>(rec.seq_len, 9999);
						R.max_len = std::max(R.max_len, len); R.min_len = std::min(R.min_len, len); R.sum_len += len;
						if (c % 1000 == 0 && c < 10000000) R.sample.push_back(Read{std::string(rec.name, rec.name_len), std::string(rec.seq, rec.seq_len), std::string()});
					}
				}
			}
			at = nx;
		}
		if (at != end) { R.bad = true; R.bad_at = at; }
	};
	pool.parallel_for(T, [&](int lo, int hi) { for (int r = lo; r < hi; ++r) walk(r, false); }, 1);
	for (int r = 0; r < T; ++r) {
		if (rg[r].bad) { ix.ok = false; ix.bad_at = rg[r].bad_at; return; }
		if (!rg[r].len_err.empty() && ix.length_error.empty()) ix.length_error = rg[r].len_err;
		rg[r].rec_base = ix.n_records; rg[r].ne_base = ix.n_nonempty;
		ix.n_records += rg[r].n_rec; ix.n_nonempty += rg[r].n_ne;
	}
	ix.step = step;
	ix.sub.assign((ix.n_records + (size_t) step - 1) / (size_t) step + 1, f.n);
	pool.parallel_for(T, [&](int lo, int hi) { for (int r = lo; r < hi; ++r) walk(r, true); }, 1);
	for (int r = 0; r < T; ++r) {
		ix.max_len = std::max(ix.max_len, rg[r].max_len); ix.min_len = std::min(ix.min_len, rg[r].min_len); ix.sum_len += rg[r].sum_len;
		for (Read &x : rg[r].sample) ix.sample.push_back(std::move(x));
	}
	ix.count = std::min<size_t>(ix.n_nonempty, 10000001);
	ix.ok = true;
}

template <typename T>
class BoundedQueue {
public:
	explicit BoundedQueue(size_t cap) : cap_(cap) {}
	void push(T v) {
		std::unique_lock<std::mutex> lk(mu_);
		cv_push_.wait(lk, [&] { return q_.size() < cap_; });
		q_.push_back(std::move(v));
		cv_pop_.notify_one();
	}
	bool pop(T &v) {  // false: closed and drained
		std::unique_lock<std::mutex> lk(mu_);
		cv_pop_.wait(lk, [&] { return !q_.empty() || closed_; });
		if (q_.empty()) return false;
		v = std::move(q_.front());
		q_.pop_front();
		cv_push_.notify_one();
		return true;
	}
	void close() { { std::lock_guard<std::mutex> lk(mu_); closed_ = true; } cv_pop_.notify_all(); }
private:
	std::mutex mu_;
	std::condition_variable cv_push_, cv_pop_;
	std::deque<T> q_;
	size_t cap_;
	bool closed_ = false;
};

// ---- SAM text, appended to a std::string (no stdio in the hot loop) -----------------------------------------------
inline void put_u64(std::string &s, unsigned long long v) {
	char b[24];
	int i = 24;
	do { b[--i] = (char) ('0' + v % 10); v /= 10; } while (v);
	s.append(b + i, 24 - i);
}
inline void put_i64(std::string &s, long long v) { if (v < 0) { s.push_back('-'); put_u64(s, (unsigned long long) (-(v + 1)) + 1ull); } else put_u64(s, (unsigned long long) v); }
// printf("%g") of roundf(identity * 10000) / 10000 (SAMWriter.cpp:163): at most four decimals, trailing zeros dropped
inline void put_identity(std::string &s, float identity) {
	const float r = roundf(identity * 10000.0f);
	if (!(r >= 0.0f && r <= 10000.0f)) { char b[32]; const int k = snprintf(b, sizeof(b), "%g", r / 10000.0f); s.append(b, k); return; }
	const int iv = (int) r;
	if (iv == 10000) { s.push_back('1'); return; }
	if (iv == 0) { s.push_back('0'); return; }
	char b[6] = {'0', '.', (char) ('0' + iv / 1000), (char) ('0' + iv / 100 % 10), (char) ('0' + iv / 10 % 10), (char) ('0' + iv % 10)};
	int k = 6;
	while (b[k - 1] == '0') --k;
	s.append(b, k);
}

// NGM_HIP_PROFILE=file: a sampling profile of the host side without any tool installed -- ITIMER_PROF ticks (1 kHz of process
// CPU time, delivered to whichever thread is running), the interrupted instruction's address per tick, written as
// "module offset count" lines at exit (resolve with addr2line -e <module> -f -C -i <offset>).
namespace prof {
constexpr int kSlots = 1 << 16;
std::atomic<uintptr_t> g_ip[kSlots];
std::atomic<unsigned> g_n{0};
void on_tick(int, siginfo_t *, void *uc) {
	const unsigned at = g_n.fetch_add(1, std::memory_order_relaxed);
	if (at < (unsigned) kSlots) g_ip[at].store((uintptr_t) ((ucontext_t *) uc)->uc_mcontext.gregs[REG_RIP], std::memory_order_relaxed);
}
void start() {
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_tick;
	sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, nullptr);
	struct itimerval tv = {{0, 1000}, {0, 1000}};
	setitimer(ITIMER_PROF, &tv, nullptr);
}
void dump(const char *path) {
	struct itimerval off = {{0, 0}, {0, 0}};
	setitimer(ITIMER_PROF, &off, nullptr);
	const unsigned n = std::min(g_n.load(), (unsigned) kSlots);
	std::map<std::pair<std::string, uintptr_t>, unsigned> hist;
	for (unsigned i = 0; i < n; ++i) {
		const uintptr_t ip = g_ip[i].load();
		Dl_info di;
		if (dladdr((void *) ip, &di) && di.dli_fname) hist[{di.dli_fname, ip - (uintptr_t) di.dli_fbase}] += 1;
		else hist[{"?", ip}] += 1;
	}
	if (FILE *f = fopen(path, "w")) {
		fprintf(f, "# %u samples (1 ms of process CPU time each)\n", g_n.load());
		for (const auto &e : hist) fprintf(f, "%s 0x%lx %u\n", e.first.first.c_str(), (unsigned long) e.first.second, e.second);
		fclose(f);
	}
}
}  // namespace prof

}  // namespace

// `-g a,b,... --shard-output`: one process per GPU.  Child k runs this program again with `--device <k-th GPU> --shard
// k/N -o <out>.shard<k>`: its own reference copy in that GPU's HBM, its own reader, workers, paired-end state and writer -- nothing is
// shared, so N GPUs write N files at once instead of queueing behind one writer.  Shard 0 carries the header; the pieces are
// appended to <out> in shard order (in-kernel copies) unless --keep-shards asks for `cat <out>.shard*` to be left to the caller.
int run_sharded(int argc, char **argv, const Opts &o) {
	const int N = (int) o.devices.size();
	std::vector<pid_t> kids;
	std::vector<std::string> parts;
	for (int k = 0; k < N; ++k) parts.push_back(o.out + ".shard" + std::to_string(k));
	// cold cache: ONE process builds the index and writes the cache files (this program with the reference only), then the shards load
	// them -- not N builds of the same index, N times the host memory, all writing the same two files
	{
		const std::string enc = o.ref + "-enc.2.ngm", ht = o.ref + "-ht-" + std::to_string(o.kmer) + "-" + std::to_string(o.bs_mapping ? 0 : o.kmer_skip) + ".3.ngm";
		// (a reference directory that cannot be written gets no cache either way: every shard then builds its own index in memory, and
		// a build in front of them would only be one more)
		std::string dir = o.ref;
		const size_t slash = dir.find_last_of('/');
		dir = slash == std::string::npos ? "." : slash == 0 ? "/" : dir.substr(0, slash);
		if (!o.skip_save && access(dir.c_str(), W_OK) == 0 && (access(enc.c_str(), R_OK) != 0 || access(ht.c_str(), R_OK) != 0)) {
			info("MAIN", "No index cache next to the reference: building it once before the shard processes start");
			const pid_t pid = fork();
			if (pid < 0) die("fork failed");
			if (pid == 0) {
				std::vector<std::string> a = {argv[0], "-r", o.ref, "-k", std::to_string(o.kmer), "--kmer-skip", std::to_string(o.kmer_skip), "--device", std::to_string(o.devices[0])};
				if (o.bs_mapping) a.push_back("--bs-mapping");
				std::vector<char *> av;
				for (std::string &x : a) av.push_back(&x[0]);
				av.push_back(nullptr);
				execv("/proc/self/exe", av.data());
				_exit(127);
			}
			int st = 0;
			if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0) die("building the index failed (its messages are above)");
		}
	}
	// The one collective of the path: every shard's int64[8] statistics, summed.  The shards hand their vectors to this
	// process over a pipe each -- that sum is printed in any case -- and, on distinct GPUs, also run the ncclAllReduce among themselves
	// (RCCL over xGMI; the communicator id made here travels in their environment).
	bool distinct = true;
	for (int a = 0; a < N; ++a) for (int b = a + 1; b < N; ++b) if (o.devices[a] == o.devices[b]) distinct = false;
	if (distinct && !getenv("NGM_HIP_NO_RCCL")) {
		char hex[257];
		if (ngm_stats_unique_id(hex) == 0) setenv("NGM_HIP_RCCL_ID", hex, 1);
		else info("MAIN", std::string("note: no RCCL all-reduce of the statistics (") + ngm_pipeline_last_error() + "); summing the shards' vectors here");
	} else unsetenv("NGM_HIP_RCCL_ID");
	std::vector<int> stat_rd(N, -1);
	for (int k = 0; k < N; ++k) {
		int pfd[2];
		if (pipe(pfd) != 0) die("pipe failed");
		stat_rd[k] = pfd[0];
		const pid_t pid = fork();
		if (pid < 0) die("fork failed");
		if (pid == 0) {
			for (int j = 0; j <= k; ++j) close(stat_rd[j]);
			std::vector<std::string> a(argv, argv + argc);
			a.push_back("--stats-fd"); a.push_back(std::to_string(pfd[1]));
			a.push_back("--device"); a.push_back(std::to_string(o.devices[k]));
			a.push_back("--shard"); a.push_back(std::to_string(k) + "/" + std::to_string(N));
			a.push_back("-o"); a.push_back(parts[k]);
			std::vector<char *> av;
			for (std::string &x : a) av.push_back(&x[0]);
			av.push_back(nullptr);
			execv("/proc/self/exe", av.data());
			_exit(127);
		}
		close(pfd[1]);
		kids.push_back(pid);
	}
	// The shards' vectors (64 bytes each, written when a shard is done) and their exit codes, as they come.  A shard that fails must not
	// leave the others -- and this process -- waiting: on distinct GPUs they sit in ncclCommInitRank / ncclAllReduce, which have no timeout
	// and wait for EVERY rank.  So the pipes are polled, the children reaped as they exit, and the first failure ends the rest.
	long long total[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int got = 0;
	bool ok = true;
	{
		std::vector<int64_t> vec((size_t) N * 8, 0);
		std::vector<size_t> have(N, 0);
		std::vector<bool> open_fd(N, true), alive(N, true);
		int n_alive = N, n_open = N;
		auto end_the_others = [&] { for (int k = 0; k < N; ++k) if (alive[k]) kill(kids[k], SIGTERM); };
		while (n_alive > 0 || n_open > 0) {
			std::vector<struct pollfd> pf;
			std::vector<int> which;
			for (int k = 0; k < N; ++k) if (open_fd[k]) { struct pollfd x; x.fd = stat_rd[k]; x.events = POLLIN; x.revents = 0; pf.push_back(x); which.push_back(k); }
			if (!pf.empty()) (void) poll(pf.data(), (nfds_t) pf.size(), 200);
			else usleep(20000);
			for (size_t j = 0; j < pf.size(); ++j) {
				if (!(pf[j].revents & (POLLIN | POLLHUP | POLLERR))) continue;
				const int k = which[j];
				const ssize_t r = have[k] < 64 ? read(stat_rd[k], (char *) &vec[(size_t) k * 8] + have[k], 64 - have[k]) : 0;
				if (r > 0) have[k] += (size_t) r;
				if (r <= 0 || have[k] == 64) {   // (end of file, or the vector is complete)
					if (r <= 0 || (pf[j].revents & POLLHUP)) { close(stat_rd[k]); open_fd[k] = false; --n_open; }
				}
			}
			for (;;) {
				int st = 0;
				const pid_t pid = waitpid(-1, &st, WNOHANG);
				if (pid <= 0) break;
				for (int k = 0; k < N; ++k) if (alive[k] && kids[k] == pid) {
					alive[k] = false; --n_alive;
					if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { if (ok) end_the_others(); ok = false; }
				}
			}
			if (n_alive == 0) for (int k = 0; k < N; ++k) if (open_fd[k]) {   // (every writer has gone: what is in the pipes is all there will be)
				const ssize_t r = have[k] < 64 ? read(stat_rd[k], (char *) &vec[(size_t) k * 8] + have[k], 64 - have[k]) : 0;
				if (r > 0) have[k] += (size_t) r;
				close(stat_rd[k]); open_fd[k] = false; --n_open;
			}
		}
		for (int k = 0; k < N; ++k) if (have[k] == 64) { ++got; for (int j = 0; j < 8; ++j) total[j] += vec[(size_t) k * 8 + j]; }
	}
	if (!ok) die("a shard process failed (its messages are above)");
	if (got == N) {
		char dm[400];
		snprintf(dm, sizeof(dm), "Done, %d shards summed (%lld reads mapped (%.2f%%), %lld reads not mapped, %lld lines written; %lld reads; %lld pairs with both mates mapped, %lld of them broken, mean insert size %.1f)",
				N, total[1], total[0] ? 100.0 * total[1] / total[0] : 0.0, total[2], total[3], total[0], total[4], total[5], total[7] > 0 ? (double) total[6] / (double) total[7] : 0.0);
		info("MAIN", dm);
	} else info("MAIN", "warning: not every shard handed over its statistics");
	if (o.keep_shards) { info("MAIN", "Shards written: " + parts[0] + " .. " + parts.back() + " (concatenate in this order)"); return 0; }
	const auto t0 = std::chrono::steady_clock::now();
	if (rename(parts[0].c_str(), o.out.c_str()) != 0) die("cannot rename " + parts[0]);
	const int out_fd = ::open(o.out.c_str(), O_WRONLY);   // (not O_APPEND: sendfile refuses such a target)
	if (out_fd < 0 || lseek(out_fd, 0, SEEK_END) < 0) die("cannot append to " + o.out);
	for (int k = 1; k < N; ++k) {
		const int in_fd = ::open(parts[k].c_str(), O_RDONLY);
		struct stat st;
		if (in_fd < 0 || fstat(in_fd, &st) != 0) die("cannot read " + parts[k]);
		off_t left = st.st_size;
		while (left > 0) {
			const ssize_t w = sendfile(out_fd, in_fd, nullptr, (size_t) std::min<off_t>(left, (off_t) 1 << 30));
			if (w <= 0) die("write error on " + o.out);
			left -= w;
		}
		close(in_fd);
		unlink(parts[k].c_str());
	}
	if (close(out_fd) != 0) die("write error on " + o.out);
	char msg[200];
	snprintf(msg, sizeof(msg), "%d shards appended to the output in %.3f s", N, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	info("MAIN", msg);
	return 0;
}

int main(int argc, char **argv) {

	// every batch allocates and frees a few hundred MB in MB-sized pieces from ~64 threads (output chunks, record views): with
	// glibc's defaults each of them is an mmap / page-fault / munmap cycle, and the kernel's address-space lock serialises the
	// formatter threads (measured: 55 ms per 256 k reads instead of ~2).  Keep that memory in the heap.
	mallopt(M_MMAP_THRESHOLD, 32 << 20);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	mallopt(M_TOP_PAD, 64 << 20);
	const auto t_process = std::chrono::steady_clock::now();
	Opts o = parse(argc, argv);
	if (o.shard_output && (o.devices.size() > 1 || getenv("NGM_HIP_SHARD_SINGLE")) && o.shard_n == 1 && o.stats_fd < 0 && !o.out.empty() && !(o.qry.empty() && o.qry1.empty())) return run_sharded(argc, argv, o);   // (a shard process has its parent's --stats-fd)
	// a shard process of `-g a,b,... --shard-output` on distinct GPUs: join the communicator of the one collective of the path (the final
	// statistics all-reduce, RCCL over xGMI) NOW, on a thread of its own -- ncclCommInitRank takes about a second, the load of the index hides it
	ngm_stats_comm *stats_comm = nullptr;
	std::string stats_comm_error;
	std::thread stats_comm_thread;
	if (const char *id = getenv("NGM_HIP_RCCL_ID")) {
		if (o.shard_n >= 1 && o.stats_fd >= 0) {
			const std::string id_s = id;
			stats_comm_thread = std::thread([&, id_s] {
				stats_comm = ngm_stats_comm_create(o.device, o.shard_i, o.shard_n, id_s.c_str());
				if (!stats_comm) stats_comm_error = ngm_pipeline_last_error();
			});
		}
	}
	// bisulfite mapping: the index holds every reference k-mer, the run's kmer_skip applies to the reads (src/PrefixTable.cpp:199-207, src/CS.cpp:556-560)
	ngm_ref_params rp{o.kmer, o.bs_mapping ? 0 : o.kmer_skip, o.bin_size};
	info("MAIN", "NextGenMap-compatible HIP backend (gfx950)");
	if (o.bs_mapping) info("MAIN", "BS mapping enabled. Max. number of A/T per k-mer set to " + std::to_string(o.bs_cutoff));
	// one GPU: keep every host thread (this one, the workers, the pool) on the socket the GPU hangs on -- the parse / select /
	// format stages run twice as fast there as spread over both sockets of the host
	if (o.devices.size() == 1) {
		const int cpus = ngm_host_pin_to_device_node(o.device);
		if (cpus > 0) info("MAIN", "Host threads pinned to the " + std::to_string(cpus) + " CPUs of the GPU's NUMA node");
	}
	// Page-locked memory is slow to get (~0.1 s per 150 MB, ~0.9 GB for two workers): it is requested NOW, by threads of its own, under the
	// load of the index -- measured (4 M reads, MI355X box): requested after the sensitivity estimate it is 0.14 s on the input-to-output
	// path, requested under the estimate it slows that estimate's driver calls from 0.04 to 0.14 s.  Its size depends on the longest read,
	// which only the scan of the input knows: the first records give a guess, and buffers that turn out too small are replaced where
	// they are used.
	struct TextBuf { char *p; size_t cap; };
	struct EarlyPinned {
		struct W { char *rows = nullptr, *qrows = nullptr, *names = nullptr; ngm_sam_read *meta = nullptr; char *bam_raw = nullptr, *bam_out = nullptr; };
		std::thread th;
		int q = 0, batch = 0;
		bool bam = false;
		size_t bam_raw_cap = 0, bam_out_cap = 0;
		std::vector<TextBuf> text;
		std::vector<W> w;
		~EarlyPinned() { if (th.joinable()) th.join(); }
	} early;
	{
		const int early_topn = o.paired ? 1 : o.topn;
		const bool early_gpu_bam = o.bam && !o.slam_seq && !getenv("NGM_HIP_BAM_ZLIB") && !getenv("NGM_HIP_BAM_HOST_RECORDS");
		const bool early_gpu_sam = (!o.bam || early_gpu_bam) && early_topn == 1 && !o.broken_pairs && !getenv("NGM_HIP_HOST_SAM");
		const bool early_gpu_bgzf = o.bam && !early_gpu_sam && !getenv("NGM_HIP_BAM_ZLIB");
		if ((early_gpu_sam || early_gpu_bgzf) && !(o.qry.empty() && o.qry1.empty()) && !o.out.empty()) {
			size_t peek_max = 0;
			if (o.max_read_length > 0) peek_max = (size_t) o.max_read_length;
			else {
				SeqReader peek((o.qry1.empty() ? o.qry : o.qry1).c_str());
				Read r;
				for (int i = 0; i < 256 && peek.ok() && peek.next(r); ++i) peek_max = std::max(peek_max, std::min<size_t>(r.seq.size(), 9999));
			}
			if (peek_max > 0) {
				early.q = std::min(1000, (int) ((peek_max | 1) + 1));
				early.batch = o.paired ? (o.batch & ~1) : o.batch;
				early.bam = early_gpu_bgzf;
				if (!early.bam) early.text.assign(o.devices.size() * (size_t) o.workers + 2, TextBuf{nullptr, 0});
				early.w.resize(o.devices.size() * (size_t) o.workers);
				// (--bam: a batch's records -- 36 bytes + name + CIGAR + 1.5 bytes per base + tags each -- and its BGZF blocks)
				early.bam_raw_cap = (size_t) early.batch * ((size_t) 2 * early.q + 224) + (1u << 20);
				early.bam_out_cap = ngm_bgzf_bound(early.bam_raw_cap);
				early.th = std::thread([&early] {
					const size_t cap = (size_t) early.batch * ((size_t) 2 * early.q + 288) + (1u << 20);
					std::vector<std::thread> alloc;
					for (TextBuf &t : early.text) alloc.emplace_back([&t, cap] { t.p = (char *) ngm_host_alloc(cap); t.cap = t.p ? cap : 0; });
					if (early.bam) for (EarlyPinned::W &w : early.w) {
						alloc.emplace_back([&w, &early] { w.bam_raw = (char *) ngm_host_alloc(early.bam_raw_cap); });
						alloc.emplace_back([&w, &early] { w.bam_out = (char *) ngm_host_alloc(early.bam_out_cap); });
					}
					else for (EarlyPinned::W &w : early.w) alloc.emplace_back([&w, &early] {
						const size_t rows = (size_t) early.batch * early.q;
						w.rows = (char *) ngm_host_alloc(rows); w.qrows = (char *) ngm_host_alloc(rows);
						w.meta = (ngm_sam_read *) ngm_host_alloc((size_t) early.batch * sizeof(ngm_sam_read));
						w.names = (char *) ngm_host_alloc((size_t) early.batch * 32);
					});
					for (auto &t : alloc) t.join();
				});
			}
		}
	}
	// an index cache next to the FASTA is loaded instead of rebuilding; a fresh build is saved for the next run unless
	// --skip-save (src/PrefixTable.cpp:232-262, SequenceProvider.cpp:264-330)
	const std::string ht_cache = o.ref + "-ht-" + std::to_string(o.kmer) + "-" + std::to_string(rp.kmer_skip) + ".3.ngm";
	ngm_ref *ref = ngm_ref_create_from_fasta(o.device, &rp, o.ref.c_str());
	if (!ref) die(ngm_pipeline_last_error());
	const bool had_cache = ngm_ref_loaded_from_cache(ref) != 0;  // (an unreadable or corrupt cache was rebuilt and is rewritten below)
	if (had_cache) info("PREPROCESS", "Reading reference index from " + ht_cache);
	else if (!o.skip_save) {
		if (ngm_ref_write_ngm_cache(ref, o.ref.c_str()) < 0) info("PREPROCESS", std::string("could not save the index: ") + ngm_pipeline_last_error());
		else info("PREPROCESS", "Writing reference index to " + ht_cache);
	}
	// the layout the candidate search gathers from (canonical pair buckets) belongs to the preparation of the reference, not to the first batch
	if (ngm_ref_prepare_search(ref, o.bs_mapping) < 0) die(ngm_pipeline_last_error());
	info("PREPROCESS", "index entries: " + std::to_string(ngm_ref_index_entries(ref)) + ", max. k-mer frequency " +
			std::to_string(o.max_kfreq > 0 ? o.max_kfreq : ngm_ref_auto_max_kfreq(ref)));
	{
		char tmsg[160];
		snprintf(tmsg, sizeof(tmsg), "Reference and index ready: %.3f s (%s)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_process).count(),
				had_cache ? "loaded from the cache files" : "built");
		info("PREPROCESS", tmsg);
	}
	const std::string first_input = o.qry1.empty() ? o.qry : o.qry1;  // parser1: what the estimation pass reads (ReadProvider.cpp:201)
	if (first_input.empty()) { ngm_ref_destroy(ref); return 0; }  // index only, like `ngm -r ref.fa`
	if (o.out.empty()) die("no output file given (-o/--output)");

	// ---- pass 1: read lengths + the sample for the sensitivity estimate (ReadProvider.cpp:201-305) ----------
	const auto t_input = std::chrono::steady_clock::now();  // the first input byte is read below
	const bool interleaved = o.paired && !o.qry.empty();  // ReadProvider::GenerateRead (ReadProvider.cpp:526-584)
	const std::string path0 = o.paired ? (interleaved ? o.qry : o.qry1) : o.qry, path1 = (o.paired && !interleaved) ? o.qry2 : std::string();
	MappedFile mf0, mf1;
	bool plain = false;
	if (!o.serial_reader) {
		bool ok1 = true;
		std::thread second;   // (a .gz pair inflates both files at the same time)
		if (!path1.empty()) second = std::thread([&] { ok1 = mf1.open(path1.c_str()) && mf1.plain_fastq(); });
		plain = mf0.open(path0.c_str()) && mf0.plain_fastq();
		if (second.joinable()) second.join();
		plain = plain && ok1;
		if (!plain) { mf0.release(); mf1.release(); }   // (an inflated copy the serial reader has no use for)
	}
	const int batch_reads = o.paired ? (o.batch & ~1) : o.batch;
	// the splitter hands out whole sub-ranges of sub_step records (per file): the largest power of two up to kSub that divides a batch's share
	const int per_file_reads = path1.empty() ? batch_reads : batch_reads / 2;
	int sub_step = kSub;
	while (sub_step > 1 && per_file_reads % sub_step != 0) sub_step >>= 1;
	FastqIndex ix0, ix1;
	size_t max_len = 0, min_len = 9999999, sum_len = 0, count = 0;
	std::vector<Read> sample;
	{
		// reads without a sequence are not counted (parseRead returns 0 for them: the `if (l > 0)` of ReadProvider.cpp:236)
		bool finish = false;
		auto account = [&](size_t seq_len) -> bool {  // true: keep this read for the sensitivity sample
			if (seq_len == 0) return false;
			const size_t len = std::min<size_t>(seq_len, 9999);
			max_len = std::max(max_len, len); min_len = std::min(min_len, len); sum_len += len;
			++count;
			if (count % 1000 == 0 && count < 10000000) return true;
			if (count == 10000001) { if (max_len - min_len >= 10) max_len = (size_t) (max_len * 1.1f); finish = true; }
			return false;
		};
		bool plain_ok = plain;
		if (plain_ok) {
			// both passes over a plain input are one parallel scan: record offsets for the splitter, lengths and the sample for the estimates
			mf0.prefault(); mf1.prefault();
			build_fastq_index(mf0, sub_step, true, ix0);
			if (ix0.ok && mf1.p) build_fastq_index(mf1, sub_step, false, ix1);
			if (!ix0.ok || (mf1.p && !ix1.ok)) {
				// not a strict 4-line record (multi-line sequences, stray blank lines): kseq reads those, so does the serial reader
				info("INPUT", "Record at byte " + std::to_string(!ix0.ok ? ix0.bad_at : ix1.bad_at) + " of " + (!ix0.ok ? path0 : path1) + " is not a 4-line FASTQ record: using the serial reader");
				plain_ok = plain = false; o.serial_reader = 1;
				mf0.release(); mf1.release();
			} else {
				const std::string &le = !ix0.length_error.empty() ? ix0.length_error : ix1.length_error;
				if (!le.empty()) die("Error while parsing read: sequence and quality lengths differ (" + le + ")");
				max_len = ix0.max_len; min_len = ix0.min_len; sum_len = ix0.sum_len; count = ix0.count;
				if (ix0.n_nonempty >= 10000001 && max_len - min_len >= 10) max_len = (size_t) (max_len * 1.1f);
				sample = std::move(ix0.sample);
			}
		}
		if (!plain_ok) {
			SeqReader in(first_input.c_str());
			if (!in.ok()) die("cannot open " + first_input);
			Read r;
			while (!finish && in.next(r)) if (account(r.seq.size())) sample.push_back(r);
		}
	}
	if (count == 0) die("No reads found in input file.");
	if (o.max_read_length > 0) max_len = (size_t) o.max_read_length;
	int q = (int) ((max_len | 1) + 1);
	const int avg_len = (int) (sum_len / count);
	if (q > 1000) q = 1000;
	const int corridor = o.corridor > 0 ? o.corridor : (int) (5 + avg_len * 0.15);
	char msg[512];
	snprintf(msg, sizeof(msg), "Average read length: %d (min: %zu, max: %d)", avg_len, min_len, q);
	info("INPUT", msg);
	info("INPUT", "Corridor width: " + std::to_string(corridor));

	ngm_mapper_params mp{};
	mp.qry_max_len = q; mp.corridor = corridor; mp.match_bonus = o.match; mp.mismatch_penalty = o.mismatch;
	mp.gap_read_penalty = o.gap_read; mp.gap_ref_penalty = o.gap_ref; mp.mode = o.mode; mp.variant = o.variant;
	mp.sensitivity = 0.5f; mp.kmer_min = o.kmer_min; mp.max_cmrs = o.max_cmrs; mp.max_kfreq = o.max_kfreq;
	mp.hard_clip = o.hard_clip; mp.silent_clip = o.silent_clip;
	mp.personality = o.affine ? NGM_PERSONALITY_AFFINE : NGM_PERSONALITY_LINEAR; mp.gap_extend_penalty = o.gap_extend;
	mp.min_insert_size = o.min_insert; mp.max_insert_size = o.max_insert; mp.pair_score_cutoff = 0.9f;
	mp.topn = o.topn; mp.strata = o.strata;
	mp.bs_mapping = o.bs_mapping; mp.bs_cutoff = o.bs_cutoff; mp.bs_read_skip = o.kmer_skip; mp.match_bonus_tt = o.match_tt; mp.match_bonus_tc = o.match_tc; mp.slam_seq = o.slam_seq;
	if (o.paired) info("INPUT", "Input is paired end data.");

	const auto t_indexed = std::chrono::steady_clock::now();
	// ---- sensitivity (ReadProvider.cpp:310-385) -----------------------------------------------------------
	// The estimate of the reference runs even when -s sets the value (its result is then only logged, ReadProvider.cpp:359-365): in that
	// case it runs beside the set-up of the mappers instead of in front of it.
	std::thread estimate_thread;
	struct JoinEstimate { std::thread &t; ~JoinEstimate() { if (t.joinable()) t.join(); } } join_estimate{estimate_thread};
	// (on its own thread the estimate never ends the process: its value is only logged there, so a failure is a note and the run goes on)
	auto run_estimate = [&, mp](float &sens, bool &estimated, bool logged_only) {
		char msg[512];
		auto give_up = [&](const char *why) {
			if (!logged_only) die(why);
			info("INPUT", std::string("note: no sensitivity estimate (it is only logged with -s): ") + why);
		};
		ngm_mapper_params ep = mp;
		ep.sensitivity = 0.0f;
		ep.slam_seq &= ~4;   // (the estimate looks the read k-mers up as they are: ReadProvider's own PrefixSearch, src/ReadProvider.cpp:79-124)
		const auto te0 = std::chrono::steady_clock::now();
		ngm_mapper *em = ngm_mapper_create(ref, &ep);
		if (!em) { give_up(ngm_pipeline_last_error()); return; }
		const auto te1 = std::chrono::steady_clock::now();
		const int ns = (int) sample.size();
		std::vector<char> rows((size_t) ns * q);
		for (int i = 0; i < ns; ++i) pack_row(sample[i], q, &rows[(size_t) i * q]);
		std::vector<uint32_t> offs(ns + 1);
		std::vector<float> mv(ns), both(ns);
		if (ngm_mapper_cs(em, ns, rows.data(), offs.data(), mv.data()) < 0 || ngm_mapper_cs_max_combined(em, both.data()) < 0) {
			const std::string why = ngm_pipeline_last_error();
			ngm_mapper_destroy(em);
			give_up(why.c_str());
			return;
		}
		float sum = 0.f;
		int n_used = 0;
		const int skip = o.kmer_skip + 1;
		for (int i = 0; i < ns; ++i) {
			const int L = (int) strnlen(&rows[(size_t) i * q], q);
			const int mx = (int) ceil((L - o.kmer + 1) / skip * 1.0);
			if (mx > 1.0f && both[i] <= mx) { sum += both[i] / mx; ++n_used; }
		}
		const auto te2 = std::chrono::steady_clock::now();
		ngm_mapper_destroy(em);
		if (getenv("NGM_HIP_HOST_TIMING")) {
			char tm[200];
			snprintf(tm, sizeof(tm), "Sensitivity estimate, s: mapper (+ search layout of the index) %.3f | candidate search of %d reads %.3f | mapper released %.3f",
					std::chrono::duration<double>(te1 - te0).count(), ns, std::chrono::duration<double>(te2 - te1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - te2).count());
			info("INPUT", tm);
		}
		{
			// ReadProvider.cpp:324-352; with no usable sample read the average is 0/0 = NaN and std::max(0.3f, NaN) is 0.3
			const float avg = sum / n_used * 1.0f;
			sens = std::min(std::max(0.3f, avg), 0.9f);
			snprintf(msg, sizeof(msg), "Estimated sensitivity: %f", sens);
			info("INPUT", msg);
			float modifier = 0;
			if (o.very_fast) modifier = 0.7f * (1.0f - sens);
			else if (o.fast) modifier = 0.35f * (1.0f - sens);
			else if (o.very_sensitive) modifier = -0.7f * sens;
			else if (o.sensitive) modifier = -0.35f * sens;
			sens += modifier;
			estimated = true;
			// the reference hands the estimate to the search through its string-typed configuration: Override("sensitivity",
			// float) prints it with "%f" (src/config/Config.cpp:185-189) and CS reads it back with atof -- six decimals.  On a
			// threshold like 29 votes x 0.7241379 = 21 that rounding decides whether a 21-vote region is a candidate (found by
			// the drop-in run of the real program, tests/test_gpu_dropin.py)
			{ char b[32]; snprintf(b, sizeof(b), "%f", sens); sens = (float) atof(b); }
		}
	};
	float sens = 0.5f;
	bool estimated = false;
	if (o.bs_mapping) {
		if (o.sensitivity < 0) info("INPUT", "Sensitivity parameter set to 0.5");   // ReadProvider.cpp:317, :378-386: no estimate in this mode
		estimated = true;
	} else if (count >= 1000 && !sample.empty()) {
		if (o.sensitivity >= 0) estimate_thread = std::thread([&] { float s2 = 0.5f; bool e2 = false; run_estimate(s2, e2, true); });
		else run_estimate(sens, estimated, false);
	}
	if (o.sensitivity >= 0) sens = o.sensitivity;
	else if (!estimated) info("INPUT", "Sensitivity parameter neither set nor estimated. Falling back to default.");
	mp.sensitivity = sens;


	const auto t_estimated = std::chrono::steady_clock::now();
	// ---- output ------------------------------------------------------------------------------------------------------
	// the output is written by the writer thread alone, in input order.  Measured alternatives (round 2, 10 M reads, 4.2 GB of
	// SAM): pwrite from the pool threads -- the writes serialise on the file's inode lock and the threads queueing there are
	// missing from the pool (3.5 M reads/s with 65 chunks per batch, 2.8 M and 32 s of system time with 500); a shared file
	// mapping of each batch's range filled by the pool -- 1.6 M reads/s, the page faults cost more than the lock.
	const int out_fd = ::open(o.out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (out_fd < 0) die("cannot write " + o.out);
	uint64_t out_off = 0;
	auto put_all = [&](const char *p, size_t n, uint64_t off) -> bool {
		while (n) {
			// (4 MB pieces: 13.9 GB/s into the page cache against 9-11 GB/s for one call per 32 MB and more, profiles/r03_write_calibration.txt)
			const ssize_t w = pwrite(out_fd, p, std::min<size_t>(n, (size_t) 4 << 20), (off_t) off);
			if (w <= 0) return false;
			p += w; n -= (size_t) w; off += (uint64_t) w;
		}
		return true;
	};
	std::vector<std::string> contig_names;
	std::vector<uint64_t> contig_lens;
	for (int i = 0; i < ngm_ref_contig_count(ref); ++i) { contig_names.push_back(ngm_ref_contig_name(ref, i)); contig_lens.push_back(ngm_ref_contig_len(ref, i)); }
	{
		static const char *tag[12] = {"ID", "CN", "DS", "DT", "FO", "KS", "LB", "PG", "PI", "PL", "PU", "SM"};
		std::string h = "@HD\tVN:1.0\tSO:unsorted\n";
		std::string rg;
		if (!o.rg[0].empty()) {  // SAMWriter.cpp:46-80
			rg = "@RG\tID:" + o.rg[0];
			for (int t = 1; t < 12; ++t) if (!o.rg[t].empty()) { rg += "\t"; rg += tag[t]; rg += ":" + o.rg[t]; }
			rg += "\n";
		}
		if (!o.bam) {
			for (size_t i = 0; i < contig_names.size(); ++i) { h += "@SQ\tSN:" + contig_names[i] + "\tLN:"; put_u64(h, contig_lens[i]); h += "\n"; }
			h += "@PG\tID:ngm\tPN:ngm\tVN:0.5.5-hip\tCL:\"" + o.cmdline + "\"\n";
			h += rg;
			if (o.shard_i == 0) {   // (--shard i/N: the header travels with the first shard)
				if (!put_all(h.data(), h.size(), out_off)) die("write error on " + o.out);
				out_off += h.size();
			}
		} else {
			// BAMWriter::DoWriteProlog (BAMWriter.cpp:18-110) through bamtools' SamFormatPrinter: @HD, @RG, @PG (ID PN CL VN); the
			// contigs only in the binary dictionary
			h += rg;
			h += "@PG\tID:ngm\tPN:ngm\tCL:\"" + o.cmdline + "\"\tVN:0.5.5-hip\n";
			std::string raw, z;
			ngm::bam::put_header(raw, h, contig_names, contig_lens);
			if (!ngm::bam::bgzf_compress(raw.data(), raw.size(), z)) die("BGZF compression failed");
			if (o.shard_i == 0) {
				if (!put_all(z.data(), z.size(), out_off)) die("write error on " + o.out);
				out_off += z.size();
			}
		}
	}
	const std::string rg_mapped = o.rg[0].empty() ? std::string() : "RG:Z:" + o.rg[0] + "\t";
	const std::string rg_unmapped = o.rg[0].empty() ? std::string() : "\tRG:Z:" + o.rg[0];
	const size_t stride = (size_t) 4 * q;
	const int max_insert = o.max_insert > 0 ? o.max_insert : 2147483647;
	const int topn = o.paired ? 1 : o.topn;

	// ---- the references / mappers: one reference per GPU (the first one exists), `workers` mappers per GPU ---------------
	std::vector<ngm_ref *> refs(1, ref);
	for (size_t d = 1; d < o.devices.size(); ++d) {
		ngm_ref *r2 = ngm_ref_create_from_fasta(o.devices[d], &rp, o.ref.c_str());  // the cache written above loads in seconds
		if (!r2 || ngm_ref_prepare_search(r2, o.bs_mapping) < 0) die(ngm_pipeline_last_error());
		refs.push_back(r2);
	}
	ngm_pair_state *pair_state = ngm_pair_state_create();
	// SAM text on the GPU (csrc/sam_device.h) for plain SAM output with one alignment per read; BAM and -n > 1 are formatted here
	// --bam: the records and their BGZF blocks are written by the GPU as well (sam_device.h's BAM mode + bgzf_device.h); with -n > 1,
	// --broken-pairs or SLAM-seq tags the records are formatted here and only the blocks come from the GPU
	// (NGM_HIP_BAM_HOST_RECORDS=1 forces that; NGM_HIP_BAM_ZLIB=1: records here, zlib level 6 on the pool -- the round-3 path)
	const bool gpu_bam = o.bam && !o.slam_seq && !getenv("NGM_HIP_BAM_ZLIB") && !getenv("NGM_HIP_BAM_HOST_RECORDS");
	const bool gpu_sam = (!o.bam || gpu_bam) && topn == 1 && !o.broken_pairs && !getenv("NGM_HIP_HOST_SAM");
	const bool gpu_bgzf = o.bam && !gpu_sam && !getenv("NGM_HIP_BAM_ZLIB");
	std::atomic<long long> t_bgzf_gpu_us{0}, t_bgzf_call_us{0};
	std::atomic<unsigned long long> bgzf_in_bytes{0}, bgzf_out_bytes{0};
	struct Worker { ngm_mapper *m = nullptr; char *rows = nullptr; size_t rows_cap = 0; std::vector<ngm_hit> hits; std::vector<char> cig, md;
		char *qrows = nullptr, *names = nullptr; size_t names_cap = 0; ngm_sam_read *meta = nullptr;
		ngm_bgzf *bz = nullptr; char *bam_raw = nullptr, *bam_out = nullptr; size_t bam_raw_cap = 0, bam_out_cap = 0; };
	std::vector<Worker> workers(o.devices.size() * (size_t) o.workers);
	for (size_t w = 0; w < workers.size(); ++w) {
		workers[w].m = ngm_mapper_create(refs[w % refs.size()], &mp);
		if (!workers[w].m) die(ngm_pipeline_last_error());
		ngm_mapper_set_pair_state(workers[w].m, pair_state);
		ngm_mapper_set_fast_pairing(workers[w].m, o.fast_pairing);
		if (gpu_sam) {
			ngm_sam_options so{};
			so.paired = o.paired; so.min_insert_size = o.min_insert; so.max_insert_size = o.max_insert; so.min_mq = o.min_mq;
			so.min_identity = o.min_identity; so.min_residues = o.min_residues; so.no_unal = o.no_unal; so.rg_id = o.rg[0].empty() ? nullptr : o.rg[0].c_str(); so.bs_mapping = o.bs_mapping; so.slam_seq = o.slam_seq; so.bam = o.bam ? 1 : 0;
			if (ngm_mapper_set_sam_options(workers[w].m, &so) < 0) die(ngm_pipeline_last_error());
		}
		ngm_mapper_set_reference_cs_batch(workers[w].m, 1800000 / std::max(1, avg_len));
		ngm_mapper_set_reference_score_buffer(workers[w].m, o.ref_score_buffer);
		if (gpu_bgzf) {
			workers[w].bz = ngm_bgzf_create(o.devices[w % o.devices.size()]);
			if (!workers[w].bz) die(ngm_pipeline_last_error());
		}
	}

	// ---- the record -> SAM line code (SAMWriter::DoWriteReadGeneric, SAMWriter.cpp:98-228) -----------------------------
	struct View { const Rec *r; const ngm_hit *h; const char *row; int L; const char *cigar, *md; };
	auto passes = [&](const View &v) {  // GenericReadWriter.h:205-215, :262-273
		float min_res = o.min_residues;
		if (min_res <= 1.0f) min_res = v.L * min_res;
		return v.h->mapped && v.h->mapq >= o.min_mq && v.h->identity >= o.min_identity && (float) (v.L - v.h->qstart - v.h->qend) >= min_res;
	};
	// SLAM-seq tags (SAMWriter.cpp:203-221, GenericReadWriter::computeSlaSeqTags over the per-column records of computeCigarMD,
	// SWOclCigar.cpp:484-540); the device twin is sam_slam_tags (csrc/sam_device.h)
	auto slam_tags = [&](std::string &s, const View &v) {
		const ngm_hit &h = *v.h;
		const int L = v.L;
		auto read_char = [&](int i) -> char {
			if (!h.reverse) return v.row[i];
			const char ch = v.row[L - 1 - i];
			return ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
		};
		auto read_class = [](char ch) -> unsigned { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 5u : 4u; };
		int span = 0;
		for (const char *c = v.cigar, *e = c + strlen(c); c < e;) { int num = 0; while (c < e && *c >= '0' && *c <= '9') num = num * 10 + (*c++ - '0'); if (c < e) { if (*c == 'M' || *c == 'D') span += num; ++c; } }
		std::vector<uint8_t> refc((size_t) span + 1, 5);
		(void) ngm_ref_host_classes(ref, ngm_ref_contig_start(ref, h.contig) + h.pos, span, refc.data());
		int rates[25] = {0};
		std::string mp;
		int read_i = h.qstart, ref_i = 0;
		const bool variant_cpu = o.variant == NGM_VARIANT_OCL_CPU, alt_tables = (o.slam_seq & 2) != 0;
		for (const char *c = v.cigar, *e = c + strlen(c); c < e;) {
			int num = 0;
			while (c < e && *c >= '0' && *c <= '9') num = num * 10 + (*c++ - '0');
			if (c >= e) break;
			const char op = *c++;
			if (op == 'M') {
				for (int k2 = 0; k2 < num; ++k2) {
					const unsigned fc = refc[ref_i + k2];
					const char rch = read_char(read_i + k2);
					const unsigned rc = read_class(rch);
					const int type = 5 * (int) (fc <= 3u ? fc : 4u) + (int) (rc <= 3u ? rc : 4u);
					rates[type] += 1;
					const char fch = fc == 0u ? 'A' : fc == 1u ? 'C' : fc == 2u ? 'G' : fc == 3u ? 'T' : fc == 4u ? 'x' : 'N';
					const bool eq = !variant_cpu ? (rch == fch) : (alt_tables ? rc == fc : (rc <= 3u && rc == fc));
					if (!eq) { if (!mp.empty()) mp.push_back(','); put_i64(mp, type); mp.push_back(':'); put_i64(mp, read_i + k2 + 1); mp.push_back(':'); put_i64(mp, ref_i + k2 + 1); }
				}
				read_i += num; ref_i += num;
			} else if (op == 'I') read_i += num;
			else if (op == 'D') ref_i += num;
		}
		s += "\tTC:i:"; put_i64(s, h.reverse ? rates[0 * 5 + 2] : rates[3 * 5 + 1]);
		s += "\tRA:Z:";
		for (int i = 0; i < 25; ++i) { if (i) s.push_back(','); put_i64(s, rates[i]); }
		if (!mp.empty()) { s += "\tMP:Z:"; s += mp; }
	};
	struct BamMate { int ref; long long pos0; long long tlen; };  // what BAMWriter::DoWritePair passes on (0-based, -1 = none; its own TLEN rule)
	auto write_mapped = [&](std::string &s, size_t &n_written, const View &v, int flags, const char *rnext, unsigned long long pnext, long long tlen, const BamMate &bm) {
		const ngm_hit &h = *v.h;
		const int L = v.L;
		const bool noq = v.r->qual_len == 0;
		if (h.reverse) flags |= 0x10;
		const bool clip = o.hard_clip || o.silent_clip;
		const int s0 = clip ? h.qstart : 0, sl = clip ? L - h.qstart - h.qend : L;
		if (o.bam) {  // BAMWriter::DoWriteReadGeneric (BAMWriter.cpp:147-298)
			char seq[1024], qual[1024];
			const int n = std::max(0, std::min(sl, 1000));
			const int QL = std::min<int>((int) v.r->qual_len, L);
			for (int t = 0; t < n; ++t) {
				if (!h.reverse) { seq[t] = v.row[s0 + t]; qual[t] = (s0 + t < QL) ? v.r->qual[s0 + t] : ':'; }
				else {
					const char ch = v.row[L - 1 - (s0 + t)];
					seq[t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
					qual[t] = (QL - 1 - (s0 + t) >= 0) ? v.r->qual[QL - 1 - (s0 + t)] : ':';
				}
			}
			ngm::bam::Tags tg;
			tg.add_int("AS", (int) h.score); tg.add_int("NM", h.nm); tg.add_int("NH", h.n_best);
			if (o.bs_mapping) { const bool second = o.paired && (flags & 0x80); const char *zs = second ? (h.reverse ? "+-" : "--") : (h.reverse ? "-+" : "++"); tg.add_string("ZS", zs, 2); }  // BAMWriter.cpp:240-254
			tg.add_float("XI", roundf(h.identity * 10000.0f) / 10000.0f);
			tg.add_int("X0", h.n_best); tg.add_int("XE", (int) h.max_votes); tg.add_int("XR", L - h.qstart - h.qend);
			tg.add_string("MD", v.md, strlen(v.md));
			if (!o.rg[0].empty()) tg.add_string("RG", o.rg[0].data(), o.rg[0].size());
			if (o.slam_seq) {
				// BAMWriter.cpp:273-292: TC:i, RA:Z (the 25 counts, every one followed by a comma -- the SAM writer drops the last comma, this one
				// does not), MP:Z when there are mismatches
				std::string t;
				slam_tags(t, v);   // "\tTC:i:<n>\tRA:Z:<..>[\tMP:Z:<..>]"
				const size_t a_ra = t.find("\tRA:Z:"), a_mp = t.find("\tMP:Z:");
				tg.add_int("TC", atoi(t.c_str() + 6));
				const std::string ra = t.substr(a_ra + 6, (a_mp == std::string::npos ? t.size() : a_mp) - (a_ra + 6)) + ",";
				tg.add_string("RA", ra.data(), ra.size());
				if (a_mp != std::string::npos) tg.add_string("MP", t.data() + a_mp + 6, t.size() - (a_mp + 6));
			}
			ngm::bam::put_record(s, v.r->name, v.r->name_len, (uint32_t) flags, h.contig, (int) h.pos, h.mapq, v.cigar, seq, (size_t) n, noq ? nullptr : qual,
					bm.ref, (int) bm.pos0, (int) bm.tlen, tg);
			++n_written;
			return;
		}
		s.append(v.r->name, v.r->name_len); s.push_back('\t'); put_u64(s, (unsigned) flags); s.push_back('\t');
		s += contig_names[h.contig]; s.push_back('\t'); put_u64(s, (unsigned long long) h.pos + 1); s.push_back('\t'); put_i64(s, h.mapq); s.push_back('\t');
		s += v.cigar; s.push_back('\t'); s += rnext; s.push_back('\t'); put_u64(s, pnext); s.push_back('\t'); put_i64(s, tlen); s.push_back('\t');
		const size_t at = s.size();
		if (sl > 0) {
			s.resize(at + sl);
			if (!h.reverse) memcpy(&s[at], v.row + s0, sl);
			else for (int t = 0; t < sl; ++t) { const char ch = v.row[L - 1 - (s0 + t)]; s[at + t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch; }
		}
		s.push_back('\t');
		if (noq) s.push_back('*');
		else if (sl > 0) {
			// the quality string as the reference holds it: the first L characters (IParser.h copies qry_max_len - 1 at most)
			const int QL = std::min<int>((int) v.r->qual_len, L);
			const size_t aq = s.size();
			const int take = std::max(0, std::min(sl, QL - s0));
			s.resize(aq + take);
			if (!h.reverse) memcpy(&s[aq], v.r->qual + s0, take);
			else for (int t = 0; t < take; ++t) s[aq + t] = v.r->qual[QL - 1 - (s0 + t)];
		}
		s.push_back('\t');
		s += rg_mapped;
		s += "AS:i:"; put_i64(s, (int) h.score); s += "\tNM:i:"; put_i64(s, h.nm); s += "\tNH:i:"; put_i64(s, h.n_best);
		if (o.bs_mapping) {  // SAMWriter.cpp:173-187: first mates / single reads "++" or "-+", second mates "--" or "+-"
			const bool second = o.paired && (flags & 0x80);
			s += second ? (h.reverse ? "\tZS:Z:+-" : "\tZS:Z:--") : (h.reverse ? "\tZS:Z:-+" : "\tZS:Z:++");
		}
		s += "\tXI:f:"; put_identity(s, h.identity);
		s += "\tX0:i:"; put_i64(s, h.n_best); s += "\tXE:i:"; put_i64(s, (int) h.max_votes); s += "\tXR:i:"; put_i64(s, L - h.qstart - h.qend); s += "\tMD:Z:"; s += v.md;
		if (o.slam_seq) slam_tags(s, v);
		s.push_back('\n');
		++n_written;
	};
	// SAMWriter::DoWriteUnmappedReadGeneric (SAMWriter.cpp:311-372): contig < 0 prints '*'
	auto write_unmapped = [&](std::string &s, size_t &n_written, const View &v, int flags, int contig, unsigned long long pos1, char rnext, unsigned long long pnext1) {
		if (o.no_unal) return;
		const bool noq = v.r->qual_len == 0;
		if (o.bam) {  // BAMWriter::DoWriteUnmappedReadGeneric (BAMWriter.cpp:300-375): reference / mate fields = the mapped mate's, 0-based
			char qual[1024];
			const int n = std::min(v.L, 1000), QL = std::min<int>((int) v.r->qual_len, v.L);
			for (int t = 0; t < n; ++t) qual[t] = t < QL ? v.r->qual[t] : ':';
			ngm::bam::Tags tg;
			if (!o.rg[0].empty()) tg.add_string("RG", o.rg[0].data(), o.rg[0].size());
			const int p0 = contig >= 0 ? (int) pos1 - 1 : -1;
			ngm::bam::put_record(s, v.r->name, v.r->name_len, (uint32_t) (flags | 0x4), contig >= 0 ? contig : -1, p0, 0, nullptr, v.row, (size_t) n,
					noq ? nullptr : qual, contig >= 0 ? contig : -1, p0, 0, tg);
			++n_written;
			return;
		}
		s.append(v.r->name, v.r->name_len); s.push_back('\t'); put_u64(s, (unsigned) (flags | 0x4)); s.push_back('\t');
		if (contig >= 0) s += contig_names[contig]; else s.push_back('*');
		s.push_back('\t'); put_u64(s, pos1); s += "\t0\t*\t"; s.push_back(rnext); s.push_back('\t'); put_u64(s, pnext1); s += "\t0\t";
		s.append(v.row, v.L); s.push_back('\t');
		if (noq) s.push_back('*'); else s.append(v.r->qual, std::min<int>((int) v.r->qual_len, v.L));
		s += rg_unmapped;
		s.push_back('\n');
		++n_written;
	};
	// one worker's batch, records [lo, hi) (paired: lo and hi even) -> SAM text + counters
	// AlignmentBuffer::WriteRead's pair counters (src/AlignmentBuffer.cpp:175-199): pairs with both mates mapped, those of them broken, the
	// others' insert sizes -- summed over the run (the GPU formatter counts the same: ngm_mapper_last_pair_stats)
	std::atomic<uint64_t> pair_stat[3];
	for (auto &x : pair_stat) x = 0;
	auto format_range = [&](const Batch &b, const Worker &w, int lo, int hi, std::string &s, size_t &n_total, size_t &n_mapped, size_t &n_written) {
		struct PairAcc { std::atomic<uint64_t> *t; uint64_t v[3] = {0, 0, 0}; ~PairAcc() { for (int k = 0; k < 3; ++k) if (v[k]) t[k] += v[k]; } } pacc{pair_stat};
		auto view = [&](int i, int t = 0) {
			const size_t e = (size_t) i * topn + t;
			View v{&b.recs[i], &w.hits[e], w.rows + (size_t) i * q, 0, &w.cig[e * stride], &w.md[e * stride]};
			v.L = (int) strnlen(v.row, q);
			return v;
		};
		s.reserve((size_t) (hi - lo) * (size_t) (2 * q + 160));
		if (!o.paired) {
			std::vector<std::tuple<int, unsigned long long, int>> seen;
			for (int i = lo; i < hi; ++i) {
				const View v = view(i);
				if (v.r->seq_len == 0) continue;  // NGMNames::Empty reads are discarded (GenericReadWriter.h:245-247)
				++n_total;
				if (topn == 1) {
					if (!passes(v)) { write_unmapped(s, n_written, v, 0, -1, 0, '*', 0); continue; }
					++n_mapped;
					write_mapped(s, n_written, v, 0, "*", 0, 0, BamMate{-1, -1, 0});
					continue;
				}
				// GenericReadWriter::WriteRead with several alignments (GenericReadWriter.h:199-243) behind AlignmentBuffer::WriteRead
				// (AlignmentBuffer.cpp:165-175), as the reference does it -- quirks included, because on a repeat-rich genome they decide records
				// (tests/test_gpu_humanlike.py): `mapped` starts as the answer of the LAST alignment's coordinate conversion ("TODO: fix for
				// -n > 1" there), and the identity / residue filter is sticky: once an alignment fails it, no later one is written either.
				// One record per distinct location, 0x100 on all but candidate 0.
				const ngm_hit &h0 = *v.h;
				const int calc = o.strata ? (h0.n_best <= topn ? h0.n_best : 0) : std::min(h0.n_candidates, topn);   // read->Calculated (ScoreBuffer::topNSE)
				bool mapped = calc > 0 && h0.mapq >= o.min_mq && view(i, calc - 1).h->mapped;   // (AlignmentBuffer.cpp:47: the read's MAPQ against min_mq)
				bool once = false;
				seen.clear();
				float min_res = o.min_residues;
				if (min_res <= 1.0f) min_res = v.L * min_res;
				for (int t = 0; t < calc && mapped; ++t) {
					const View vt = view(i, t);
					mapped = vt.h->identity >= o.min_identity && (float) (vt.L - vt.h->qstart - vt.h->qend) >= min_res;
					if (!mapped) break;
					once = true;
					if (!vt.h->mapped) continue;   // (its position did not convert: the reference would print an unconverted location here -- not mirrored)
					const auto key = std::make_tuple(vt.h->contig, (unsigned long long) vt.h->pos, vt.h->reverse);
					if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
					seen.push_back(key);
					write_mapped(s, n_written, vt, t ? 0x100 : 0, "*", 0, 0, BamMate{-1, -1, 0});
				}
				if (once) ++n_mapped; else write_unmapped(s, n_written, v, 0, -1, 0, '*', 0);
			}
			return;
		}
		for (int i = lo; i + 1 < hi; i += 2) {
			// read1 = the first mate (even ReadId), written second by AlignmentBuffer::WriteRead; read2 = its mate
			const View v1 = view(i), v2 = view(i + 1);
			if (v2.r->unit & 2) {
				// --broken-pairs: a read without its mate goes out like a single-end read (AlignmentBuffer.cpp:201-203, GenericReadWriter::WriteRead)
				if (v1.r->seq_len == 0) continue;
				++n_total;
				if (!passes(v1)) { write_unmapped(s, n_written, v1, 0, -1, 0, '*', 0); continue; }
				++n_mapped;
				write_mapped(s, n_written, v1, 0, "*", 0, 0, BamMate{-1, -1, 0});
				continue;
			}
			if (v1.r->seq_len == 0 || v2.r->seq_len == 0) continue;  // GenericReadWriter.h:250-252
			const ngm_hit &h1 = *v1.h, &h2 = *v2.h;
			if ((h1.pair_flags | h2.pair_flags) & NGM_PAIR_LOST) continue;   // the reference never writes this pair (ngm_mapper_set_reference_score_buffer)
			n_total += 2;
			// AlignmentBuffer::WriteRead (AlignmentBuffer.cpp:175-199): is the pair consistent?
			bool paired_fail = (h1.pair_flags & NGM_PAIR_FAILED) || (h2.pair_flags & NGM_PAIR_FAILED);
			if (h1.mapped && h2.mapped) {
				const long long distance = (h2.pos > h1.pos) ? (long long) (h2.pos - h1.pos) + v1.L : (long long) (h1.pos - h2.pos) + v2.L;
				++pacc.v[0];
				if (h1.contig != h2.contig || distance < o.min_insert || distance > max_insert || h1.reverse == h2.reverse) { paired_fail = true; ++pacc.v[1]; }
				else pacc.v[2] += (uint64_t) distance;
			}
			const bool m1 = passes(v1), m2 = passes(v2);  // GenericReadWriter::WritePair
			n_mapped += (m1 ? 1 : 0) + (m2 ? 1 : 0);
			const bool swp = (v1.r->unit & 4) != 0;   // (--broken-pairs: the reference's read ids have lost their parity, see the reader)
			const int f1 = 0x1 | (swp ? 0x80 : 0x40), f2 = 0x1 | (swp ? 0x40 : 0x80);  // SAMWriter::DoWritePair (SAMWriter.cpp:230-310)
			const unsigned long long p1 = h1.pos + 1, p2 = h2.pos + 1;
			if (!m1 && !m2) {
				write_unmapped(s, n_written, v2, f2 | 0x8, -1, 0, '*', 0);
				write_unmapped(s, n_written, v1, f1 | 0x8, -1, 0, '*', 0);
			} else if (!m1) {
				write_mapped(s, n_written, v2, f2 | 0x8, "=", p2, 0, BamMate{h2.contig, (long long) h2.pos, 0});
				write_unmapped(s, n_written, v1, f1, h2.contig, p2, '=', p2);
			} else if (!m2) {
				write_unmapped(s, n_written, v2, f2, h1.contig, p1, '=', p1);
				write_mapped(s, n_written, v1, f1 | 0x8, "=", p1, 0, BamMate{h1.contig, (long long) h1.pos, 0});
			} else if (!paired_fail) {
				if (!h1.reverse) {
					const long long d = ((long long) h2.pos + v2.L - h2.qstart - h2.qend) - (long long) h1.pos;
					const long long db = (long long) h2.pos + v2.L - (long long) h1.pos;  // BAMWriter.cpp:428-433: the whole read length
					write_mapped(s, n_written, v2, f2 | 0x2, "=", p1, -d, BamMate{h2.contig, (long long) h1.pos, -db});
					write_mapped(s, n_written, v1, f1 | 0x2 | 0x20, "=", p2, d, BamMate{h2.contig, (long long) h2.pos, db});
				} else if (!h2.reverse) {
					const long long d = ((long long) h1.pos + v1.L - h1.qstart - h1.qend) - (long long) h2.pos;
					const long long db = (long long) h1.pos + v1.L - (long long) h2.pos;
					write_mapped(s, n_written, v2, f2 | 0x2 | 0x20, "=", p1, d, BamMate{h2.contig, (long long) h1.pos, db});
					write_mapped(s, n_written, v1, f1 | 0x2, "=", p2, -d, BamMate{h2.contig, (long long) h2.pos, -db});
				}
			} else {
				write_mapped(s, n_written, v2, f2 | (h1.reverse ? 0x20 : 0), contig_names[h1.contig].c_str(), p1, 0, BamMate{h1.contig, (long long) h1.pos, 0});
				write_mapped(s, n_written, v1, f1 | (h2.reverse ? 0x20 : 0), contig_names[h2.contig].c_str(), p2, 0, BamMate{h2.contig, (long long) h2.pos, 0});
			}
		}
	};

	// ---- the pipeline ---------------------------------------------------------------------------------------------
	ngm::ThreadPool &pool = ngm::ThreadPool::instance();
	BoundedQueue<std::unique_ptr<Batch>> q_in(workers.size() + 2);
	std::mutex out_mu;
	std::condition_variable out_cv;
	std::map<uint64_t, std::unique_ptr<Batch>> out_ready;
	std::atomic<uint64_t> next_write{0};   // the batch the writer waits for (written under out_mu)
	bool workers_done = false;
	std::atomic<bool> failed{false};
	std::string fail_msg;
	std::mutex fail_mu;
	std::mutex text_mu;
	std::condition_variable text_cv;
	// (every waiter looks at `failed`: wake them all -- ADVICE r3)
	auto fail = [&](const std::string &m2) {
		{ std::lock_guard<std::mutex> lk(fail_mu); if (!failed.exchange(true)) fail_msg = m2; }
		{ std::lock_guard<std::mutex> lk(out_mu); }
		out_cv.notify_all();
		{ std::lock_guard<std::mutex> lk(text_mu); }
		text_cv.notify_all();
	};

	auto strip_mate = [&](const char *name, uint32_t &len) {  // ReadProvider::NextRead (ReadProvider.cpp:419-422)
		if (len >= 2 && name[len - 2] == o.pe_delimiter) len -= 2;
	};

	std::thread splitter([&] {
		uint64_t seq = 0;
		if (plain) {
			// record boundaries: the index of the estimation pass (every sub_step-th record offset per file, so that a batch is parsed in parallel)
			const bool two = !path1.empty();
			const char *uneven = "Error in input file. Number of reads in input not even. Please check the input or mapped in single-end mode.";
			if (two ? ix0.n_records != ix1.n_records : (o.paired && (ix0.n_records & 1))) fail(uneven);
			auto take = [&](const FastqIndex &ix, size_t r0, int cnt, std::vector<size_t> &sub) {
				const size_t s0 = r0 / (size_t) sub_step, s1 = (r0 + (size_t) cnt + (size_t) sub_step - 1) / (size_t) sub_step;
				sub.assign(ix.sub.begin() + (long) s0, ix.sub.begin() + (long) s1 + 1);
			};
			// --shard i/N: records [lo, hi) of each file, the boundaries on whole sub-ranges (and whole pairs of an interleaved file)
			const size_t gran = (size_t) std::max(sub_step, (o.paired && !two) ? 2 : 1);
			auto bound = [&](int i) -> size_t { return i >= o.shard_n ? ix0.n_records : (size_t) ((unsigned __int128) ix0.n_records * (unsigned) i / (unsigned) o.shard_n) / gran * gran; };
			const size_t rec_lo = bound(o.shard_i), rec_hi = bound(o.shard_i + 1);
			// (a quarter-size first batch per worker, so that the writer starts earlier, was measured slower: mapping pass 0.57 / 0.64 s against 0.52 / 0.54 s)
			const int first_div = 1;
			const size_t first_share = std::max<size_t>((size_t) sub_step, (size_t) per_file_reads / (size_t) first_div / (size_t) sub_step * (size_t) sub_step);
			for (size_t r0 = rec_lo, step = 0; !failed && r0 < rec_hi; r0 += step) {
				auto b = std::make_unique<Batch>();
				step = seq < workers.size() ? first_share : (size_t) per_file_reads;
				b->seq = seq++;
				b->n0 = (int) std::min<size_t>(step, rec_hi - r0);
				take(ix0, r0, b->n0, b->sub0);
				if (two) { b->n1 = b->n0; take(ix1, r0, b->n1, b->sub1); }
				b->n = b->n0 + b->n1;
				q_in.push(std::move(b));
			}
		} else if (o.shard_n > 1) {
			fail("--shard needs 4-line FASTQ input (plain, or .gz small enough to be inflated into memory): the shard boundaries come from the record index");
		} else {
			SeqReader in1(path0.c_str());
			std::unique_ptr<SeqReader> in2(path1.empty() ? nullptr : new SeqReader(path1.c_str()));
			if (!in1.ok() || (in2 && !in2->ok())) { fail(o.paired ? "cannot open the paired-end input" : "cannot open " + path0); q_in.close(); return; }
			auto b = std::make_unique<Batch>();
			Read a, c, spare;
			bool have_spare = false;
			const uint64_t bp_units_per_batch = std::max<uint64_t>(1, (uint64_t) ((1800000 / std::max(1, avg_len)) & ~1) / 2);   // CS.cpp:26, :542; NGM.cpp:237-242
			uint64_t bp_units = 0, bp_reads = 0, bp_cur_start = 0;
			for (;;) {
				if (failed) break;
				if (!o.paired) {
					if (!in1.next(a)) break;
					b->owned.push_back(std::move(a));
				} else if (o.broken_pairs) {
					// ReadProvider::GenerateRead with acceptBrokenPaired (ReadProvider.cpp:529-575): two consecutive records whose names differ are
					// not mates -- the first is mapped alone (its mate's row holds a placeholder), the second becomes the first of the next pair;
					// a record left over at the end is mapped alone too.  Read ids (NGM.cpp:252-267): a batch of the reference's CS thread hands out
					// ids m_CurStart + 2 j, + 1 to its j-th pair and then advances m_CurStart by the reads it actually got -- after an odd number
					// the first mates of the following batch carry ODD ids, and SAMWriter::DoWritePair (SAMWriter.cpp:235-244) gives them 0x80
					if (have_spare) { a = std::move(spare); have_spare = false; }
					else if (!in1.next(a)) break;
					const bool hb = in1.next(c);
					uint32_t la = (uint32_t) a.name.size(), lc = (uint32_t) c.name.size();
					strip_mate(a.name.data(), la);
					if (hb) strip_mate(c.name.data(), lc);
					const bool mates = hb && la == lc && memcmp(a.name.data(), c.name.data(), la) == 0;
					const uint8_t swp = (bp_cur_start & 1) ? 4 : 0;
					if (mates) { a.unit = swp; c.unit = swp; b->owned.push_back(std::move(a)); b->owned.push_back(std::move(c)); bp_reads += 2; }
					else {
						a.unit = 1;
						b->owned.push_back(std::move(a));
						Read ph; ph.unit = 2;
						b->owned.push_back(std::move(ph));
						bp_reads += 1;
						if (hb) { spare = std::move(c); have_spare = true; }
					}
					if (++bp_units == bp_units_per_batch) { bp_cur_start += bp_reads; bp_units = 0; bp_reads = 0; }
				} else {
					const bool ha = in1.next(a), hb = ha ? (in2 ? in2->next(c) : in1.next(c)) : false;
					if (!ha) break;
					if (!hb) { fail("Error in input file. Number of reads in input not even. Please check the input or mapped in single-end mode."); break; }
					b->owned.push_back(std::move(a)); b->owned.push_back(std::move(c));
				}
				if ((int) b->owned.size() >= batch_reads) { b->seq = seq++; b->n = (int) b->owned.size(); q_in.push(std::move(b)); b = std::make_unique<Batch>(); }
			}
			if (!failed && !b->owned.empty()) { b->seq = seq++; b->n = (int) b->owned.size(); q_in.push(std::move(b)); }
		}
		q_in.close();
	});

	// The formatted text of a batch (~100 MB in MB-sized pieces) and its record views are recycled instead of freed: fresh memory
	// means page faults from 64 threads at once, and those serialise on the address-space lock.
	std::mutex spare_mu;
	std::vector<std::vector<std::string>> spare_chunks;
	std::vector<std::vector<Rec>> spare_recs;
	std::atomic<long long> t_gpu_us{0};  // HIP-event time of the mapping kernels, summed over the batches
	std::atomic<long long> t_wait_us{0}, t_parse_us{0}, t_map_us{0}, t_format_us{0}, t_format_cpu_us{0}, t_parse_cpu_us{0}, t_write_us{0}, t_write_cpu_us{0};
	auto us_since = [](std::chrono::steady_clock::time_point t0) { return (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
	// GPU-formatted text: page-locked buffers that travel worker -> writer -> pool.  Their number bounds the formatted text in
	// flight (a slow output file then stalls the workers instead of piling the output up in memory)
	std::vector<TextBuf> text_free;
	const size_t text_cap0 = (size_t) batch_reads * ((size_t) 2 * q + 288) + (1u << 20);
	if (early.th.joinable()) early.th.join();
	if (early.bam && gpu_bgzf && early.w.size() == workers.size()) {
		// (the GPU BGZF path: the buffers carry their sizes, a batch that needs more replaces them)
		for (size_t i = 0; i < workers.size(); ++i) if (early.w[i].bam_raw && early.w[i].bam_out) {
			workers[i].bam_raw = early.w[i].bam_raw; workers[i].bam_out = early.w[i].bam_out;
			workers[i].bam_raw_cap = early.bam_raw_cap; workers[i].bam_out_cap = early.bam_out_cap;
			early.w[i].bam_raw = early.w[i].bam_out = nullptr;
		}
	}
	for (EarlyPinned::W &w : early.w) { ngm_host_free(w.bam_raw); ngm_host_free(w.bam_out); w.bam_raw = w.bam_out = nullptr; }
	bool early_ok = gpu_sam && !early.bam && early.q >= q && early.batch == batch_reads && early.w.size() == workers.size();
	for (const TextBuf &t : early.text) early_ok = early_ok && t.p;
	for (const EarlyPinned::W &w : early.w) early_ok = early_ok && w.rows && w.qrows && w.meta && w.names;
	if (early_ok) {
		text_free = early.text;
		for (size_t i = 0; i < workers.size(); ++i) {
			Worker &w = workers[i];
			w.rows = early.w[i].rows; w.qrows = early.w[i].qrows; w.meta = early.w[i].meta; w.names = early.w[i].names;
			w.rows_cap = (size_t) batch_reads * early.q; w.names_cap = (size_t) batch_reads * 32;
		}
	} else {
		for (const TextBuf &t : early.text) ngm_host_free(t.p);
		for (const EarlyPinned::W &w : early.w) { ngm_host_free(w.rows); ngm_host_free(w.qrows); ngm_host_free(w.meta); ngm_host_free(w.names); }
	}
	if (gpu_sam && !early_ok) {
		// page-locked memory is slow to get (~0.1 s per 150 MB): everything a worker needs, and the text pool, at once and in parallel
		// (measured: requesting it under the sensitivity estimate slows the driver calls of that estimate down by more than it saves;
		// under the input scan -- `early` above -- it is free)
		std::vector<std::thread> alloc;
		text_free.resize(workers.size() + 2);
		for (TextBuf &t : text_free) alloc.emplace_back([&t, text_cap0] { t.p = (char *) ngm_host_alloc(text_cap0); t.cap = text_cap0; });
		for (Worker &w : workers) alloc.emplace_back([&w, batch_reads, q] {
			w.rows_cap = (size_t) batch_reads * q;
			w.rows = (char *) ngm_host_alloc(w.rows_cap); w.qrows = (char *) ngm_host_alloc(w.rows_cap);
			w.meta = (ngm_sam_read *) ngm_host_alloc((size_t) batch_reads * sizeof(ngm_sam_read));
			w.names_cap = (size_t) batch_reads * 32; w.names = (char *) ngm_host_alloc(w.names_cap);
		});
		for (auto &t : alloc) t.join();
		for (const TextBuf &t : text_free) if (!t.p) die(ngm_pipeline_last_error());
		for (const Worker &w : workers) if (!w.rows || !w.qrows || !w.meta || !w.names) die(ngm_pipeline_last_error());
	}
	std::atomic<long long> t_sam_gpu_us{0};
	const auto t_start = std::chrono::steady_clock::now();   // the mapping pass: mappers and page-locked buffers exist, the splitter has begun to cut batches
	auto worker_main = [&](Worker &w) {
		std::unique_ptr<Batch> b;
		for (;;) {
			auto tw = std::chrono::steady_clock::now();
			if (!q_in.pop(b)) break;
			t_wait_us += us_since(tw);
			auto tp = std::chrono::steady_clock::now();
			if (failed) { ngm_mapper_set_batch_seq(w.m, b->seq); if (o.paired) (void) ngm_mapper_map_pe(w.m, 0, nullptr, nullptr, nullptr, nullptr); continue; }
			const int n = b->n;
			if ((size_t) n * q > w.rows_cap) {
				ngm_host_free(w.rows);
				w.rows_cap = (size_t) std::max(n, batch_reads) * q;
				w.rows = (char *) ngm_host_alloc(w.rows_cap);
				if (!w.rows) { fail(ngm_pipeline_last_error()); w.rows_cap = 0; }
				if (gpu_sam && w.rows) {
					ngm_host_free(w.qrows); ngm_host_free(w.meta);
					w.qrows = (char *) ngm_host_alloc(w.rows_cap);
					w.meta = (ngm_sam_read *) ngm_host_alloc((size_t) std::max(n, batch_reads) * sizeof(ngm_sam_read));
					if (!w.qrows || !w.meta) { fail(ngm_pipeline_last_error()); ngm_host_free(w.rows); w.rows = nullptr; w.rows_cap = 0; }
				}
			}
			{
				std::lock_guard<std::mutex> lk(spare_mu);
				if (!spare_recs.empty()) { b->recs = std::move(spare_recs.back()); spare_recs.pop_back(); }
				if (!spare_chunks.empty()) { b->chunks = std::move(spare_chunks.back()); spare_chunks.pop_back(); }
			}
			b->recs.resize(n);
			std::atomic<bool> bad{false};
			std::string bad_msg;
			if (w.rows && !b->owned.empty()) {
				pool.parallel_for(n, [&](int lo, int hi) {
					for (int i = lo; i < hi; ++i) {
						const Read &r = b->owned[i];
						b->recs[i] = Rec{r.name.data(), r.seq.data(), r.qual.data(), (uint32_t) r.name.size(), (uint32_t) r.seq.size(), (uint32_t) r.qual.size(), r.unit};
						if (o.paired) strip_mate(b->recs[i].name, b->recs[i].name_len);
						pack_row_view(r.seq.data(), r.seq.size(), q, w.rows + (size_t) i * q);
						if (gpu_sam) memcpy(w.qrows + (size_t) i * q, r.qual.data(), std::min<size_t>(r.qual.size(), (size_t) q - 1));
					}
				}, 4096);
			} else if (w.rows) {
				// plain input: sub-range s of file f holds records [s sub_step, ...) of that file's share of the batch; record j of file f is batch record
				// j (one file) or 2 j + f (two files)
				const bool two = b->n1 > 0;
				const int nsub0 = (int) b->sub0.size() - 1, nsub1 = two ? (int) b->sub1.size() - 1 : 0;
				pool.parallel_for(nsub0 + nsub1, [&](int lo, int hi) {
					const auto t_cpu = std::chrono::steady_clock::now();
					struct Acc { std::atomic<long long> &a; std::chrono::steady_clock::time_point t; ~Acc() { a += (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); } } acc{t_parse_cpu_us, t_cpu};
					for (int sidx = lo; sidx < hi; ++sidx) {
						const int f = sidx < nsub0 ? 0 : 1, sl = f ? sidx - nsub0 : sidx;
						const MappedFile &mf = f ? mf1 : mf0;
						const std::vector<size_t> &sub = f ? b->sub1 : b->sub0;
						const int cnt = std::min(sub_step, (f ? b->n1 : b->n0) - sl * sub_step);
						size_t at = sub[sl];
						for (int j = 0; j < cnt; ++j) {
							const int i = two ? 2 * (sl * sub_step + j) + f : sl * sub_step + j;
							Rec &r = b->recs[i];
							const size_t nx = mf.record(at, r);
							if (!nx) { if (!bad.exchange(true)) bad_msg = "malformed FASTQ record at byte " + std::to_string(at); return; }
							if (r.qual_len != r.seq_len) { if (!bad.exchange(true)) bad_msg = "Error while parsing read: sequence and quality lengths differ (" + std::string(r.name, r.name_len) + ")"; return; }  // IParser.h copyToRead
							at = nx;
							if (o.paired) strip_mate(r.name, r.name_len);
							pack_row_view(r.seq, r.seq_len, q, w.rows + (size_t) i * q);
							if (gpu_sam) memcpy(w.qrows + (size_t) i * q, r.qual, std::min<size_t>(r.qual_len, (size_t) q - 1));
						}
					}
				}, 1);
			}
			if (bad) fail(bad_msg);
			if (!failed && o.paired) {
				std::atomic<int> first_bad{n};
				pool.parallel_for(n / 2, [&](int lo, int hi) {
					for (int pi = lo; pi < hi; ++pi) {
						const Rec &a = b->recs[2 * pi], &c = b->recs[2 * pi + 1];
						if (c.unit & 2) continue;   // --broken-pairs: a read without its mate
						if (a.name_len != c.name_len || memcmp(a.name, c.name, a.name_len) != 0) {
							int seen = first_bad.load();
							while (2 * pi < seen && !first_bad.compare_exchange_weak(seen, 2 * pi)) {}
							return;
						}
					}
				}, 8192);
				if (first_bad.load() < n) {
					const Rec &a = b->recs[first_bad.load()], &c = b->recs[first_bad.load() + 1];
					fail("Error while reading paired end reads. Names of mates don't match: " + std::string(a.name, a.name_len) + " and " + std::string(c.name, c.name_len) + ".");
				}
			}
			ngm_mapper_set_batch_seq(w.m, b->seq);
			if (failed) { if (o.paired) (void) ngm_mapper_map_pe(w.m, 0, nullptr, nullptr, nullptr, nullptr); continue; }
			if (gpu_sam) {
				// names: one block of bytes per batch (offsets by a prefix sum over the records), plus what the record printer needs per read
				size_t total = 0;
				for (int i = 0; i < n; ++i) {
					const Rec &r = b->recs[i];
					w.meta[i].name_off = (uint32_t) total; w.meta[i].name_len = (uint16_t) std::min<uint32_t>(r.name_len, 65535u);
					w.meta[i].qual_len = (uint16_t) (std::min<uint32_t>(r.qual_len, 0x7FFFu) | (r.seq_len == 0 ? 0x8000u : 0u));
					total += w.meta[i].name_len;
				}
				if (total + 16 > w.names_cap) {
					ngm_host_free(w.names);
					w.names_cap = std::max(total + 16, (size_t) batch_reads * 32);
					w.names = (char *) ngm_host_alloc(w.names_cap);
					if (!w.names) { fail(ngm_pipeline_last_error()); w.names_cap = 0; if (o.paired) (void) ngm_mapper_map_pe(w.m, 0, nullptr, nullptr, nullptr, nullptr); continue; }
				}
				pool.parallel_for(n, [&](int lo, int hi) { for (int i = lo; i < hi; ++i) memcpy(w.names + w.meta[i].name_off, b->recs[i].name, w.meta[i].name_len); }, 8192);
				t_parse_us += us_since(tp);
				auto tm = std::chrono::steady_clock::now();
				TextBuf tb{nullptr, 0};
				{
					// (buffers go out in no particular order and come back only when the writer has written batch `next_write`; the
					// worker that holds exactly that batch must never wait for one -- it takes a fresh buffer instead -- and nobody waits
					// once the run has failed)
					std::unique_lock<std::mutex> lk(text_mu);
					text_cv.wait(lk, [&] { return !text_free.empty() || b->seq == next_write.load() || failed.load(); });
					if (!text_free.empty()) { tb = text_free.back(); text_free.pop_back(); }
				}
				if (failed) { if (tb.p) { std::lock_guard<std::mutex> lk(text_mu); text_free.push_back(tb); } if (o.paired) (void) ngm_mapper_map_pe(w.m, 0, nullptr, nullptr, nullptr, nullptr); continue; }
				if (!tb.p) {
					tb.cap = text_cap0;
					tb.p = (char *) ngm_host_alloc(tb.cap);
					if (!tb.p) { fail(ngm_pipeline_last_error()); if (o.paired) (void) ngm_mapper_map_pe(w.m, 0, nullptr, nullptr, nullptr, nullptr); continue; }
				}
				uint64_t st[3] = {0, 0, 0};
				float sam_ms = 0.f;
				long long len = ngm_mapper_map_sam(w.m, n, w.rows, w.qrows, w.names, total, w.meta, tb.p, tb.cap, st, &sam_ms);
				if (len > (long long) tb.cap) {  // (long CIGAR / MD strings: a larger buffer for this batch)
					ngm_host_free(tb.p);
					tb.cap = (size_t) len + (1u << 20);
					tb.p = (char *) ngm_host_alloc(tb.cap);
					const int got = tb.p ? ngm_mapper_sam_fetch(w.m, tb.p, tb.cap) : -1;
					if (got < 0) len = -1; else if (o.bam) len = got;   // (BAM: the BGZF blocks are made by the fetch; it says how long they are)
				}
				if (len < 0) { fail(ngm_pipeline_last_error()); std::lock_guard<std::mutex> lk(text_mu); if (tb.p) text_free.push_back(tb); text_cv.notify_one(); continue; }
				t_map_us += us_since(tm);
				if (o.paired) { uint64_t ps3[3] = {0, 0, 0}; if (ngm_mapper_last_pair_stats(w.m, ps3) == 0) for (int k2 = 0; k2 < 3; ++k2) pair_stat[k2] += ps3[k2]; }
				{ float kms[8] = {0}; if (ngm_mapper_last_kernel_ms(w.m, kms) == 0) { double sum = sam_ms; for (int k2 = 0; k2 < 7; ++k2) sum += kms[k2]; t_gpu_us += (long long) (sum * 1000.0); } }
				t_sam_gpu_us += (long long) (sam_ms * 1000.0);
				b->text = tb.p; b->text_cap = tb.cap; b->text_len = (size_t) len;
				b->n_total = st[0]; b->n_mapped = st[1]; b->n_written = st[2];
				{
					std::lock_guard<std::mutex> lk(spare_mu);
					b->recs.clear();
					spare_recs.push_back(std::move(b->recs));
				}
				b->owned.clear(); b->owned.shrink_to_fit();
				{
					std::lock_guard<std::mutex> lk(out_mu);
					out_ready[b->seq] = std::move(b);
				}
				out_cv.notify_all();
				continue;
			}
			t_parse_us += us_since(tp);
			auto tm = std::chrono::steady_clock::now();
			w.hits.resize((size_t) n * topn); w.cig.resize((size_t) n * topn * stride); w.md.resize((size_t) n * topn * stride);
			const int rc = o.paired ? ngm_mapper_map_pe(w.m, n, w.rows, w.hits.data(), w.cig.data(), w.md.data())
			                        : ngm_mapper_map_se(w.m, n, w.rows, w.hits.data(), w.cig.data(), w.md.data());
			if (rc < 0) { fail(ngm_pipeline_last_error()); continue; }
			t_map_us += us_since(tm);
			{ float kms[8] = {0}; if (ngm_mapper_last_kernel_ms(w.m, kms) == 0) { double sum = 0; for (int k2 = 0; k2 < 7; ++k2) sum += kms[k2]; t_gpu_us += (long long) (sum * 1000.0); } }
			auto tf = std::chrono::steady_clock::now();
			// format: chunks of whole pairs
			const int units = o.paired ? n / 2 : n, per = o.paired ? 2 : 1;
			const int n_chunks = std::max(1, std::min(units / 256 + 1, pool.size() * 8));  // many more chunks than threads: the slowest thread decides when the batch is done
			b->chunks.resize(n_chunks);
			for (std::string &c : b->chunks) c.clear();  // (keeps the capacity of the batch this vector served before)
			std::vector<size_t> ct(n_chunks, 0), cm(n_chunks, 0), cw(n_chunks, 0);
			pool.parallel_for(n_chunks, [&](int lo, int hi) {
				const auto t_cpu = std::chrono::steady_clock::now();
				struct Acc { std::atomic<long long> &a; std::chrono::steady_clock::time_point t; ~Acc() { a += (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); } } acc{t_format_cpu_us, t_cpu};
				for (int c = lo; c < hi; ++c) {
					const int u0 = (int) ((long long) units * c / n_chunks), u1 = (int) ((long long) units * (c + 1) / n_chunks);
					size_t t_loc = 0, m_loc = 0, w_loc = 0;  // (not ct[c] & co. directly: neighbouring chunks share cache lines, and these are bumped per record)
					format_range(*b, w, u0 * per, u1 * per, b->chunks[c], t_loc, m_loc, w_loc);
					ct[c] = t_loc; cm[c] = m_loc; cw[c] = w_loc;
					if (o.bam && !gpu_bgzf && !b->chunks[c].empty()) {  // every chunk becomes whole BGZF blocks: they concatenate into one valid file
						std::string z;
						z.reserve(b->chunks[c].size() / 3 + 64);
						if (!ngm::bam::bgzf_compress(b->chunks[c].data(), b->chunks[c].size(), z)) fail("BGZF compression failed");
						b->chunks[c].swap(z);
					}
				}
			}, 1, (o.bam && !gpu_bgzf) ? ngm::ThreadPool::cpu_quota() : 0);   // (BAM with zlib: records + deflate keep every thread busy for the whole batch)
			for (int c = 0; c < n_chunks; ++c) { b->n_total += ct[c]; b->n_mapped += cm[c]; b->n_written += cw[c]; }
			if (gpu_bgzf) {
				// the batch's records, chunk after chunk, in page-locked memory -> whole BGZF blocks from the GPU (a batch ends with a short
				// block: batches -- and shards -- concatenate into one valid file) -> pieces for the writer
				const auto tz = std::chrono::steady_clock::now();
				std::vector<size_t> off(n_chunks + 1, 0);
				for (int c = 0; c < n_chunks; ++c) off[c + 1] = off[c] + b->chunks[c].size();
				const size_t total = off[n_chunks];
				long long zlen = 0;
				if (total > 0) {
					if (total > w.bam_raw_cap) {
						ngm_host_free(w.bam_raw); ngm_host_free(w.bam_out);
						w.bam_raw_cap = total + total / 4 + (1u << 20); w.bam_out_cap = ngm_bgzf_bound(w.bam_raw_cap);
						w.bam_raw = (char *) ngm_host_alloc(w.bam_raw_cap); w.bam_out = (char *) ngm_host_alloc(w.bam_out_cap);
						if (!w.bam_raw || !w.bam_out) { w.bam_raw_cap = w.bam_out_cap = 0; fail(ngm_pipeline_last_error()); continue; }
					}
					pool.parallel_for(n_chunks, [&](int lo, int hi) { for (int c = lo; c < hi; ++c) memcpy(w.bam_raw + off[c], b->chunks[c].data(), b->chunks[c].size()); }, 1);
					zlen = ngm_bgzf_compress(w.bz, w.bam_raw, total, w.bam_out, w.bam_out_cap);
					if (zlen < 0) { fail(ngm_pipeline_last_error()); continue; }
					t_bgzf_gpu_us += (long long) (ngm_bgzf_last_kernel_ms(w.bz) * 1000.0f);
					bgzf_in_bytes += total; bgzf_out_bytes += (unsigned long long) zlen;
				}
				const int n_out = (int) std::max<long long>(1, std::min<long long>(n_chunks, zlen / (2 << 20) + 1));
				b->chunks.resize((size_t) n_out);
				pool.parallel_for(n_out, [&](int lo, int hi) {
					for (int c = lo; c < hi; ++c) {
						const size_t a = (size_t) ((unsigned long long) zlen * (unsigned) c / (unsigned) n_out), e = (size_t) ((unsigned long long) zlen * (unsigned) (c + 1) / (unsigned) n_out);
						b->chunks[c].assign(w.bam_out + a, e - a);
					}
				}, 1);
				t_bgzf_call_us += us_since(tz);
			}
			{
				std::lock_guard<std::mutex> lk(spare_mu);
				b->recs.clear();
				spare_recs.push_back(std::move(b->recs));
			}
			b->owned.clear(); b->owned.shrink_to_fit();
			t_format_us += us_since(tf);
			{
				// back-pressure: formatted batches wait here while the writer is behind -- except the one it is waiting for
				std::unique_lock<std::mutex> lk(out_mu);
				out_cv.wait(lk, [&] { return out_ready.size() < workers.size() + 2 || b->seq == next_write.load() || failed.load(); });
				out_ready[b->seq] = std::move(b);
			}
			out_cv.notify_all();
		}
	};
	size_t n_total = 0, n_mapped = 0, n_written = 0, n_read = 0;   // (n_read: records of the batches; n_total: reads the writer counted as mapped or unmapped)
	std::thread writer([&] {
		uint64_t next = 0;
		for (;;) {
			std::unique_ptr<Batch> b;
			{
				std::unique_lock<std::mutex> lk(out_mu);
				next_write = next;
				out_cv.notify_all();
				{ std::lock_guard<std::mutex> lk2(text_mu); }   // (a worker between its predicate and its wait holds text_mu: it sees the new value or the notify)
				text_cv.notify_all();
				out_cv.wait(lk, [&] { return out_ready.count(next) || workers_done; });
				auto it = out_ready.find(next);
				if (it == out_ready.end()) return;  // workers are done and the next batch never came (failure) or everything is written
				b = std::move(it->second);
				out_ready.erase(it);
			}
			out_cv.notify_all();
			if (b->text) {
				const auto t_wr = std::chrono::steady_clock::now();
				// (the batch's text in k slices written at the same time was tried: one file takes 9-14 GB/s whatever the number of writers,
				// profiles/r03_write_calibration.txt)
				if (b->text_len && !put_all(b->text, b->text_len, out_off)) fail("write error on " + o.out);
				out_off += b->text_len;
				t_write_us += (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wr).count();
				{ std::lock_guard<std::mutex> lk(text_mu); text_free.push_back(TextBuf{b->text, b->text_cap}); }
				text_cv.notify_one();
				n_total += b->n_total; n_mapped += b->n_mapped; n_written += b->n_written; n_read += (size_t) b->n;
				++next;
				continue;
			}
			std::vector<uint64_t> offs(b->chunks.size());
			for (size_t c = 0; c < b->chunks.size(); ++c) { offs[c] = out_off; out_off += b->chunks[c].size(); }
			const auto t_wr = std::chrono::steady_clock::now();
			// one writer: buffered writes to one file serialise on its inode lock anyway, and pool threads queueing there (64 of
			// them, 500 chunks per batch) burn more system time than the copies take (measured: 32 s of it for 4.2 GB)
			for (size_t c = 0; c < b->chunks.size(); ++c) if (!b->chunks[c].empty() && !put_all(b->chunks[c].data(), b->chunks[c].size(), offs[c])) fail("write error on " + o.out);
			t_write_us += (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wr).count();
			{
				std::lock_guard<std::mutex> lk(spare_mu);
				spare_chunks.push_back(std::move(b->chunks));
			}
			n_total += b->n_total; n_mapped += b->n_mapped; n_written += b->n_written; n_read += (size_t) b->n;
			++next;
		}
	});
	if (getenv("NGM_HIP_PROFILE")) prof::start();  // the mapping pass only
	struct timespec cpu0;
	clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &cpu0);
	struct rusage ru0;
	getrusage(RUSAGE_SELF, &ru0);
	{
		std::vector<std::thread> th;
		for (Worker &w : workers) th.emplace_back([&, pw = &w] { worker_main(*pw); });
		for (auto &t : th) t.join();
	}
	splitter.join();
	if (estimate_thread.joinable()) estimate_thread.join();   // (it finished long ago unless the input is a handful of reads; the reference it uses is released below)
	{ std::lock_guard<std::mutex> lk(out_mu); workers_done = true; }
	out_cv.notify_all();
	writer.join();
	if (o.bam && o.shard_i == o.shard_n - 1) { std::string z; ngm::bam::bgzf_eof(z); if (!put_all(z.data(), z.size(), out_off)) fail("write error on " + o.out); out_off += z.size(); }   // (--shard: the end-of-file block travels with the last shard)
	if (close(out_fd) != 0) fail("write error on " + o.out);
	if (failed) die(fail_msg);
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	// src/NGM_main.cpp:150-160: reads that were read but never reached the writer's counters ("discarded": empty reads, and the pairs the
	// reference loses -- ngm_mapper_set_reference_score_buffer) are listed when there are any
	const size_t n_discarded = n_read > n_total ? n_read - n_total : 0;
	if (n_discarded) snprintf(msg, sizeof(msg), "Done (%zu reads mapped (%.2f%%), %zu reads not mapped (%zu discarded), %zu lines written)", n_mapped,
			100.0 * n_mapped / std::max<size_t>(1, n_total + n_discarded), n_total - n_mapped + n_discarded, n_discarded, n_written);
	else snprintf(msg, sizeof(msg), "Done (%zu reads mapped (%.2f%%), %zu reads not mapped, %zu lines written)", n_mapped,
			n_total ? 100.0 * n_mapped / n_total : 0.0, n_total - n_mapped, n_written);
	info("MAIN", msg);
	{
		// this process's share of the mapping statistics
		int64_t sv[8] = {(int64_t) n_read, (int64_t) n_mapped, (int64_t) (n_read - n_mapped), (int64_t) n_written, (int64_t) pair_stat[0].load(), (int64_t) pair_stat[1].load(),
				(int64_t) pair_stat[2].load(), (int64_t) (pair_stat[0].load() - pair_stat[1].load())};
		if (o.stats_fd >= 0) {   // a shard process: the parent sums the shards' vectors
			if (write(o.stats_fd, sv, sizeof(sv)) != (ssize_t) sizeof(sv)) info("MAIN", "warning: could not hand the statistics to the parent process");
			close(o.stats_fd);
		}
		if (stats_comm_thread.joinable()) stats_comm_thread.join();
		if (stats_comm) {
			const auto t_ar = std::chrono::steady_clock::now();
			int64_t all[8];
			memcpy(all, sv, sizeof(all));
			if (ngm_stats_allreduce(stats_comm, all) == 0) {
				if (o.shard_i == 0) {
					snprintf(msg, sizeof(msg), "Statistics all-reduce over %d GPUs (RCCL, %.0f us): %lld reads, %lld mapped, %lld not mapped, %lld lines written; %lld pairs with both mates mapped, %lld of them broken, mean insert size %.1f",
							o.shard_n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_ar).count(), (long long) all[0], (long long) all[1], (long long) all[2], (long long) all[3],
							(long long) all[4], (long long) all[5], all[7] > 0 ? (double) all[6] / (double) all[7] : 0.0);
					info("MAIN", msg);
				}
			} else info("MAIN", std::string("warning: statistics all-reduce failed: ") + ngm_pipeline_last_error());
			ngm_stats_comm_destroy(stats_comm);
		} else if (!stats_comm_error.empty() && o.shard_i == 0) info("MAIN", "note: no RCCL communicator for the statistics (" + stats_comm_error + "); the parent process sums them");
	}
	snprintf(msg, sizeof(msg), "Mapping pass: %.3f s, %.0f reads/s (%zu GPU(s) x %d worker(s), %d host threads, %s input)", secs, n_total / std::max(1e-9, secs),
			o.devices.size(), o.workers, pool.size(), plain ? (mf0.owned ? "gzip FASTQ inflated to memory" : "memory-mapped plain FASTQ") : "serial reader");
	info("MAIN", msg);
	snprintf(msg, sizeof(msg), "GPU kernels: %.3f s of the %.3f s mapping pass (%.0f %%; candidate search, gathers, score, select, align, traceback by HIP events)",
			t_gpu_us / 1e6, secs, 100.0 * (t_gpu_us / 1e6) / std::max(1e-9, secs));
	info("MAIN", msg);
	if (!gpu_sam) {
		snprintf(msg, sizeof(msg), "Records%s formatted on the host pool: %.3f s of thread time = %.0f %% of %d threads x %.3f s", (o.bam && !gpu_bgzf) ? " + BGZF blocks" : "", t_format_cpu_us / 1e6,
				100.0 * (t_format_cpu_us / 1e6) / std::max(1e-9, pool.size() * secs), pool.size(), secs);
		info("MAIN", msg);
	}
	if (gpu_bgzf) {
		snprintf(msg, sizeof(msg), "BGZF blocks written by the GPU: %.1f MB of records -> %.1f MB, %.3f s of kernels, %.3f s inside the calls (copies to and from page-locked memory, both transfers, the kernels)",
				bgzf_in_bytes.load() / 1e6, bgzf_out_bytes.load() / 1e6, t_bgzf_gpu_us / 1e6, t_bgzf_call_us / 1e6);
		info("MAIN", msg);
	}
	if (gpu_sam) { snprintf(msg, sizeof(msg), "%s on the GPU: %.3f s of kernels (included above)", o.bam ? "BAM records and their BGZF blocks written" : "SAM text assembled", t_sam_gpu_us / 1e6); info("MAIN", msg); }
	snprintf(msg, sizeof(msg), "Input to output: %.3f s (estimation pass + mapping pass, first input byte to output closed)",
			std::chrono::duration<double>(std::chrono::steady_clock::now() - t_input).count());
	info("MAIN", msg);
	if (getenv("NGM_HIP_HOST_TIMING")) {
		auto span = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
		snprintf(msg, sizeof(msg), "Before the mapping pass, s: input mapped + record index + read lengths %.3f | sensitivity estimate %.3f | header, mappers, page-locked buffers %.3f",
				span(t_input, t_indexed), span(t_indexed, t_estimated), span(t_estimated, t_start));
		info("MAIN", msg);
		snprintf(msg, sizeof(msg), "Worker time summed over %zu workers, s: waiting for input %.3f | parse + pack %.3f | map (GPU + library host stages) %.3f | format %.3f",
				workers.size(), t_wait_us / 1e6, t_parse_us / 1e6, t_map_us / 1e6, t_format_us / 1e6);
		info("MAIN", msg);
		{
			struct timespec cpu1;
			clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &cpu1);
			struct rusage ru1;
			getrusage(RUSAGE_SELF, &ru1);
			auto tv = [](const timeval &a, const timeval &b) { return (double) (b.tv_sec - a.tv_sec) + 1e-6 * (double) (b.tv_usec - a.tv_usec); };
			snprintf(msg, sizeof(msg), "Mapping pass, process totals: CPU %.2f s (user %.2f, system %.2f), page faults %ld minor / %ld major, context switches %ld voluntary / %ld involuntary",
					(double) (cpu1.tv_sec - cpu0.tv_sec) + 1e-9 * (double) (cpu1.tv_nsec - cpu0.tv_nsec), tv(ru0.ru_utime, ru1.ru_utime), tv(ru0.ru_stime, ru1.ru_stime),
					ru1.ru_minflt - ru0.ru_minflt, ru1.ru_majflt - ru0.ru_majflt, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw);
			info("MAIN", msg);
		}
		snprintf(msg, sizeof(msg), "Pool thread time inside the stages, s: parse + pack %.3f | format %.3f | output copies %.3f (writer wall %.3f)",
				t_parse_cpu_us / 1e6, t_format_cpu_us / 1e6, t_write_cpu_us / 1e6, t_write_us / 1e6);
		info("MAIN", msg);
	}
	{
		uint64_t pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (Worker &w : workers) { uint64_t c8[8]; if (ngm_mapper_path_counters(w.m, c8) == 0) for (int x = 0; x < 8; ++x) pc[x] += c8[x]; }
		const double nr = (double) std::max<uint64_t>(1, pc[0]);
		snprintf(msg, sizeof(msg), "Candidate search: %llu reads, %.2f candidates per read; heavy-read kernel for %llu reads (%.3f %%), exact search with the table in LDS for %llu (%.3f %%), in global memory for %llu (%.3f %%)",
				(unsigned long long) pc[0], pc[1] / nr, (unsigned long long) pc[7], 100.0 * pc[7] / nr, (unsigned long long) pc[2], 100.0 * pc[2] / nr, (unsigned long long) pc[3], 100.0 * pc[3] / nr);
		info("MAIN", msg);
		uint64_t hc[4] = {0, 0, 0, 0};
		for (Worker &w : workers) { uint64_t c4[4]; if (ngm_mapper_heavy_counters(w.m, c4) == 0) for (int x = 0; x < 4; ++x) hc[x] += c4[x]; }
		snprintf(msg, sizeof(msg), "Heavy-read kernel: %llu second passes, %llu table passes started over, %llu reads sent on to the exact kernels; table pool regrown %llu times",
				(unsigned long long) hc[0], (unsigned long long) hc[1], (unsigned long long) hc[2], (unsigned long long) hc[3]);
		info("MAIN", msg);
		uint64_t by_table = 0;
		for (Worker &w : workers) { uint64_t c2 = 0; if (ngm_mapper_order_table_reads(w.m, &c2) == 0) by_table += c2; }
		snprintf(msg, sizeof(msg), "Candidate order replay: %llu reads, %llu of them beyond the LDS replay (exact replay through buckets in global memory, %llu of them with a table there); order left undetermined for %llu reads",
				(unsigned long long) pc[4], (unsigned long long) pc[5], (unsigned long long) by_table, (unsigned long long) pc[6]);
		info("MAIN", msg);
	}
	if (o.paired && o.ref_score_buffer > 0) {
		uint64_t lost = 0;
		for (Worker &w : workers) { uint64_t c2 = 0; if (ngm_mapper_lost_pairs(w.m, &c2) == 0) lost += c2; }
		snprintf(msg, sizeof(msg), "Pairs lost as NextGenMap loses them (first mate's last score fills the %d-entry score buffer, second mate without candidates; src/ScoreBuffer.cpp:196, :519-523): %llu",
				o.ref_score_buffer, (unsigned long long) lost);
		info("MAIN", msg);
	}
	if (const char *pf = getenv("NGM_HIP_PROFILE")) prof::dump(pf);
	for (Worker &w : workers) { ngm_mapper_destroy(w.m); ngm_host_free(w.rows); ngm_host_free(w.qrows); ngm_host_free(w.names); ngm_host_free(w.meta);
		ngm_bgzf_destroy(w.bz); ngm_host_free(w.bam_raw); ngm_host_free(w.bam_out); }
	for (TextBuf &t : text_free) ngm_host_free(t.p);
	ngm_pair_state_destroy(pair_state);
	for (ngm_ref *r2 : refs) ngm_ref_destroy(r2);
	return 0;
}
