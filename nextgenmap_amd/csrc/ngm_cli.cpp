// ngm_cli.cpp -- `ngm-hip`: NextGenMap-compatible command line over the C ABI (include/ngm_pipeline.h).
//
// Mirrors, for single-end input, what ngm-core does around its hot path:
//   option names / defaults          src/config/Options.h:13-105, src/config/Config.cpp:381-557
//   read parsing                     src/parser/IParser.h:59-121 (upper-case, non-ACGT -> N, truncate to qry_max_len-1)
//   parameter estimation             src/ReadProvider.cpp:163-394 (qry_max_len, corridor, sensitivity)
//   output filter                    src/writer/GenericReadWriter.h:190-254 (min_identity, min_residues), min_mq
//   SAM header / records             src/writer/SAMWriter.cpp:17-84, :98-228, :312-373
// Everything heavy (index, candidate search, score, align) happens on the GPU behind ngm_mapper_*.
// Not supported (rejected loudly): paired-end input, --affine, BAM, bisulfite / SLAM-seq, --topn > 1, --argos, --vcf.
#include <getopt.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <unistd.h>
#include <vector>

#include "../../include/ngm_hip.h"
#include "../../include/ngm_pipeline.h"

namespace {

struct Read { std::string name, seq, qual; };

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
		if (fastq) while (r.qual.size() < r.seq.size() && getline(line)) r.qual += line;
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
	int match = 10, mismatch = 15, gap_read = -1, gap_ref = -1, gap_extend = -1, affine = 0, hard_clip = 0, silent_clip = 0, no_unal = 0, max_cmrs = 2147483647;
	int skip_save = 0;
	std::string rg[12];  // read group: ID CN DS DT FO KS LB PG PI PL PU SM (SAMWriter.cpp:46-80)
	int very_fast = 0, fast = 0, sensitive = 0, very_sensitive = 0, variant = NGM_VARIANT_OCL_GPU;
	float sensitivity = -1.f, kmer_min = 0.f, min_identity = 0.65f, min_residues = 0.5f;
	int batch = 1 << 20;
	std::string cmdline;
};

[[noreturn]] void die(const std::string &msg) { fprintf(stderr, "[ngm-hip] error: %s\n", msg.c_str()); exit(1); }
void info(const char *tag, const std::string &msg) { fprintf(stderr, "[%s] %s\n", tag, msg.c_str()); }

Opts parse(int argc, char **argv) {
	Opts o;
	for (int i = 1; i < argc; ++i) { if (i > 1) o.cmdline += " "; o.cmdline += argv[i]; }  // Config.cpp:565-574
	enum { KSKIP = 1000, HARD, SILENT, KMIN, MB, MMP, GRP, GFP, MAXCMRS, NOUNAL, NOPROG, MAXRL, BINSZ, MAXKF, VFAST, FAST, SENS, VSENS, DEVICE,
		SKIPSAVE, BATCH, VARIANT, AFFINE, GEP, PEDELIM, STRATA, RG0, RG_LAST = RG0 + 11, UNSUPPORTED };
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
		{"kernel-variant", required_argument, 0, VARIANT},
		{"qry1", required_argument, 0, '1'}, {"qry2", required_argument, 0, '2'}, {"paired", no_argument, 0, 'p'},
		{"min-insert-size", required_argument, 0, 'I'}, {"max-insert-size", required_argument, 0, 'X'}, {"pe-delimiter", required_argument, 0, PEDELIM},
		{"rg-id", required_argument, 0, RG0}, {"rg-cn", required_argument, 0, RG0 + 1}, {"rg-ds", required_argument, 0, RG0 + 2},
		{"rg-dt", required_argument, 0, RG0 + 3}, {"rg-fo", required_argument, 0, RG0 + 4}, {"rg-ks", required_argument, 0, RG0 + 5},
		{"rg-lb", required_argument, 0, RG0 + 6}, {"rg-pg", required_argument, 0, RG0 + 7}, {"rg-pi", required_argument, 0, RG0 + 8},
		{"rg-pl", required_argument, 0, RG0 + 9}, {"rg-pu", required_argument, 0, RG0 + 10}, {"rg-sm", required_argument, 0, RG0 + 11},
		{"fast-pairing", no_argument, 0, UNSUPPORTED}, {"broken-pairs", no_argument, 0, UNSUPPORTED},
		{"affine", no_argument, 0, AFFINE}, {"gap-extend-penalty", required_argument, 0, GEP}, {"bam", no_argument, 0, UNSUPPORTED}, {"bs-mapping", no_argument, 0, UNSUPPORTED},
		{"slam-seq", required_argument, 0, UNSUPPORTED}, {"topn", required_argument, 0, 'n'}, {"strata", no_argument, 0, STRATA},
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
		case 'g': break;  // the GPU is not optional here
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
		case MB: o.match = atoi(optarg); break;
		case MMP: o.mismatch = atoi(optarg); break;
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
		case DEVICE: o.device = atoi(optarg); break;
		case BATCH: o.batch = std::max(1024, atoi(optarg)); break;
		case VARIANT: o.variant = atoi(optarg) ? NGM_VARIANT_OCL_CPU : NGM_VARIANT_OCL_GPU; break;
		case UNSUPPORTED: die(std::string("option --") + lo[idx].name + " is not supported by the HIP backend yet");
		default: die("unknown option (see src/config/Options.h of NextGenMap for the option set)");
		}
	}
	if (o.ref.empty()) die("no reference given (-r/--ref)");
	if (!o.qry1.empty() && !o.qry2.empty()) o.paired = 1;  // Config.cpp:395-399
	else if (!o.qry1.empty() || !o.qry2.empty()) die("--qry1 and --qry2 must be given together");
	if (o.paired && o.topn > 1) die("Paired end mode with topn > 1 not yet supported.");  // ScoreBuffer::topNPE
	if (o.paired && o.qry.empty() && o.qry1.empty()) die("-p/--paired needs -q (interleaved mates) or --qry1/--qry2");
	// scoring defaults depend on the personality (Config.cpp:433-446)
	if (o.gap_read < 0) o.gap_read = o.affine ? 33 : 20;
	if (o.gap_ref < 0) o.gap_ref = o.affine ? 33 : 20;
	if (o.gap_extend < 0) o.gap_extend = o.affine ? 3 : 5;
	return o;
}

void pack_row(const Read &r, int q, char *row) {  // IParser.h:59-121
	memset(row, 0, q);
	if (r.seq.empty()) { row[0] = 'N'; return; }
	const int L = (int) std::min<size_t>(r.seq.size(), (size_t) q - 1);
	for (int i = 0; i < L; ++i) {
		const char c = (char) toupper((unsigned char) r.seq[i]);
		row[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
	}
}

}  // namespace

int main(int argc, char **argv) {
	Opts o = parse(argc, argv);
	ngm_ref_params rp{o.kmer, o.kmer_skip, o.bin_size};
	info("MAIN", "NextGenMap-compatible HIP backend (gfx950)");
	// an index cache next to the FASTA is loaded instead of rebuilding; a fresh build is saved for the next run unless
	// --skip-save (src/PrefixTable.cpp:232-262, SequenceProvider.cpp:264-330)
	const std::string ht_cache = o.ref + "-ht-" + std::to_string(o.kmer) + "-" + std::to_string(o.kmer_skip) + ".3.ngm";
	const bool had_cache = access(ht_cache.c_str(), R_OK) == 0 && access((o.ref + "-enc.2.ngm").c_str(), R_OK) == 0 && !getenv("NGM_HIP_NO_CACHE");
	ngm_ref *ref = ngm_ref_create_from_fasta(o.device, &rp, o.ref.c_str());
	if (!ref) die(ngm_pipeline_last_error());
	if (had_cache) info("PREPROCESS", "Reading reference index from " + ht_cache);
	else if (!o.skip_save) {
		if (ngm_ref_write_ngm_cache(ref, o.ref.c_str()) < 0) info("PREPROCESS", std::string("could not save the index: ") + ngm_pipeline_last_error());
		else info("PREPROCESS", "Writing reference index to " + ht_cache);
	}
	info("PREPROCESS", "index entries: " + std::to_string(ngm_ref_index_entries(ref)) + ", max. k-mer frequency " +
			std::to_string(o.max_kfreq > 0 ? o.max_kfreq : ngm_ref_auto_max_kfreq(ref)));
	const std::string first_input = o.qry1.empty() ? o.qry : o.qry1;  // parser1: what the estimation pass reads (ReadProvider.cpp:201)
	if (first_input.empty()) { ngm_ref_destroy(ref); return 0; }  // index only, like `ngm -r ref.fa`
	if (o.out.empty()) die("no output file given (-o/--output)");

	// ---- pass 1: read lengths + the sample for the sensitivity estimate (ReadProvider.cpp:201-305) ----------
	size_t max_len = 0, min_len = 9999999, sum_len = 0, count = 0;
	std::vector<Read> sample;
	{
		SeqReader in(first_input.c_str());
		if (!in.ok()) die("cannot open " + first_input);
		Read r;
		bool finish = false;
		while (!finish && in.next(r)) {
			const size_t len = r.seq.empty() ? 1 : std::min<size_t>(r.seq.size(), 9999);
			max_len = std::max(max_len, len); min_len = std::min(min_len, len); sum_len += len;
			++count;
			if (count % 1000 == 0 && count < 10000000) sample.push_back(r);
			else if (count == 10000001) { if (max_len - min_len >= 10) max_len = (size_t) (max_len * 1.1f); finish = true; }
		}
	}
	if (count == 0) die("No reads found in input file.");
	if (o.max_read_length > 0) max_len = (size_t) o.max_read_length;
	int q = (int) ((max_len | 1) + 1);
	const int avg_len = (int) (sum_len / count);
	if (q > 1000) q = 1000;
	const int corridor = o.corridor > 0 ? o.corridor : (int) (5 + avg_len * 0.15);
	char msg[256];
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
	if (o.paired) info("INPUT", "Input is paired end data.");

	// ---- sensitivity (ReadProvider.cpp:310-385) -----------------------------------------------------------
	float sens = 0.5f;
	bool estimated = false;
	if (count >= 1000 && !sample.empty()) {
		ngm_mapper_params ep = mp;
		ep.sensitivity = 0.0f;
		ngm_mapper *em = ngm_mapper_create(ref, &ep);
		if (!em) die(ngm_pipeline_last_error());
		const int ns = (int) sample.size();
		std::vector<char> rows((size_t) ns * q);
		for (int i = 0; i < ns; ++i) pack_row(sample[i], q, &rows[(size_t) i * q]);
		std::vector<uint32_t> offs(ns + 1);
		std::vector<float> mv(ns), both(ns);
		if (ngm_mapper_cs(em, ns, rows.data(), offs.data(), mv.data()) < 0 || ngm_mapper_cs_max_combined(em, both.data()) < 0) die(ngm_pipeline_last_error());
		float sum = 0.f;
		int n_used = 0;
		const int skip = o.kmer_skip + 1;
		for (int i = 0; i < ns; ++i) {
			const int L = (int) strnlen(&rows[(size_t) i * q], q);
			const int mx = (int) ceil((L - o.kmer + 1) / skip * 1.0);
			if (mx > 1.0f && both[i] <= mx) { sum += both[i] / mx; ++n_used; }
		}
		ngm_mapper_destroy(em);
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
		}
	}
	if (o.sensitivity >= 0) sens = o.sensitivity;
	else if (!estimated) info("INPUT", "Sensitivity parameter neither set nor estimated. Falling back to default.");
	mp.sensitivity = sens;

	ngm_mapper *m = ngm_mapper_create(ref, &mp);
	if (!m) die(ngm_pipeline_last_error());

	FILE *out = fopen(o.out.c_str(), "w");
	if (!out) die("cannot write " + o.out);
	std::vector<char> obuf(1 << 24);
	setvbuf(out, obuf.data(), _IOFBF, obuf.size());
	fprintf(out, "@HD\tVN:1.0\tSO:unsorted\n");
	for (int i = 0; i < ngm_ref_contig_count(ref); ++i)
		fprintf(out, "@SQ\tSN:%s\tLN:%llu\n", ngm_ref_contig_name(ref, i), (unsigned long long) ngm_ref_contig_len(ref, i));
	fprintf(out, "@PG\tID:ngm\tPN:ngm\tVN:0.5.5-hip\tCL:\"%s\"\n", o.cmdline.c_str());
	if (!o.rg[0].empty()) {  // SAMWriter.cpp:46-80
		static const char *tag[12] = {"ID", "CN", "DS", "DT", "FO", "KS", "LB", "PG", "PI", "PL", "PU", "SM"};
		fprintf(out, "@RG\tID:%s", o.rg[0].c_str());
		for (int t = 1; t < 12; ++t) if (!o.rg[t].empty()) fprintf(out, "\t%s:%s", tag[t], o.rg[t].c_str());
		fprintf(out, "\n");
	}
	const std::string rg_mapped = o.rg[0].empty() ? std::string() : "RG:Z:" + o.rg[0] + "\t";
	const std::string rg_unmapped = o.rg[0].empty() ? std::string() : "\tRG:Z:" + o.rg[0];

	// ---- pass 2: map in batches ------------------------------------------------------------------------------
	std::vector<Read> batch;
	std::vector<char> rows, cig, md;
	std::vector<ngm_hit> hits;
	size_t n_total = 0, n_mapped = 0, n_written = 0;
	const size_t stride = (size_t) 4 * q;
	const int max_insert = o.max_insert > 0 ? o.max_insert : 2147483647;

	struct View { const Read *r; const ngm_hit *h; const char *row; int L; const char *cigar, *md; };
	const int topn = o.paired ? 1 : o.topn;
	auto view = [&](int i, int t = 0) {
		const size_t e = (size_t) i * topn + t;
		View v{&batch[i], &hits[e], &rows[(size_t) i * q], 0, &cig[e * stride], &md[e * stride]};
		v.L = (int) strnlen(v.row, q);
		return v;
	};
	auto passes = [&](const View &v) {  // GenericReadWriter.h:205-215, :262-273
		float min_res = o.min_residues;
		if (min_res <= 1.0f) min_res = v.L * min_res;
		return v.h->mapped && v.h->mapq >= o.min_mq && v.h->identity >= o.min_identity && (float) (v.L - v.h->qstart - v.h->qend) >= min_res;
	};
	// SAMWriter::DoWriteReadGeneric (SAMWriter.cpp:98-228)
	auto write_mapped = [&](const View &v, int flags, const char *rnext, unsigned long long pnext, long long tlen) {
		const ngm_hit &h = *v.h;
		const int L = v.L;
		const bool noq = v.r->qual.empty();
		std::string seq(v.row, L), ql = noq ? std::string("*") : v.r->qual.substr(0, L);
		if (h.reverse) {
			flags |= 0x10;
			for (int t = 0; t < L; ++t) { const char ch = v.row[L - 1 - t]; seq[t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch; }
			if (!noq) std::reverse(ql.begin(), ql.end());
		}
		const bool clip = o.hard_clip || o.silent_clip;
		const int s0 = clip ? h.qstart : 0, sl = clip ? L - h.qstart - h.qend : L;
		const float identity = roundf(h.identity * 10000.0f) / 10000.0f;
		fprintf(out, "%s\t%d\t%s\t%llu\t%d\t%s\t%s\t%llu\t%lld\t%.*s\t%.*s\t%sAS:i:%d\tNM:i:%d\tNH:i:%d\tXI:f:%g\tX0:i:%d\tXE:i:%d\tXR:i:%d\tMD:Z:%s\n",
				v.r->name.c_str(), flags, ngm_ref_contig_name(ref, h.contig), (unsigned long long) h.pos + 1, h.mapq, v.cigar, rnext, pnext, tlen,
				sl, seq.c_str() + s0, noq ? 1 : sl, noq ? "*" : ql.c_str() + s0, rg_mapped.c_str(),
				(int) h.score, h.nm, h.n_best, identity, h.n_best, (int) h.max_votes, L - h.qstart - h.qend, v.md);
		++n_written;
	};
	// SAMWriter::DoWriteUnmappedReadGeneric (SAMWriter.cpp:311-372): contig < 0 prints '*'
	auto write_unmapped = [&](const View &v, int flags, int contig, unsigned long long pos1, char rnext, unsigned long long pnext1) {
		if (o.no_unal) return;
		const bool noq = v.r->qual.empty();
		const int ql = noq ? 1 : std::min<int>((int) v.r->qual.size(), v.L);
		fprintf(out, "%s\t%d\t%s\t%llu\t0\t*\t%c\t%llu\t0\t%.*s\t%.*s%s\n", v.r->name.c_str(), flags | 0x4, contig >= 0 ? ngm_ref_contig_name(ref, contig) : "*",
				pos1, rnext, pnext1, v.L, v.row, ql, noq ? "*" : v.r->qual.c_str(), rg_unmapped.c_str());
		++n_written;
	};

	auto flush = [&]() {
		const int n = (int) batch.size();
		if (n == 0) return;
		rows.assign((size_t) n * q, 0);
		for (int i = 0; i < n; ++i) pack_row(batch[i], q, &rows[(size_t) i * q]);
		hits.resize((size_t) n * topn); cig.resize((size_t) n * topn * stride); md.resize((size_t) n * topn * stride);
		const int rc = o.paired ? ngm_mapper_map_pe(m, n, rows.data(), hits.data(), cig.data(), md.data())
		                        : ngm_mapper_map_se(m, n, rows.data(), hits.data(), cig.data(), md.data());
		if (rc < 0) die(ngm_pipeline_last_error());
		if (!o.paired) {
			for (int i = 0; i < n; ++i) {
				const View v = view(i);
				if (v.r->seq.empty()) continue;  // NGMNames::Empty reads are discarded (GenericReadWriter.h:245-247)
				++n_total;
				if (topn == 1) {
					if (!passes(v)) { write_unmapped(v, 0, -1, 0, '*', 0); continue; }
					++n_mapped;
					write_mapped(v, 0, "*", 0, 0);
					continue;
				}
				// GenericReadWriter::WriteRead with several alignments (GenericReadWriter.h:199-243): every alignment that
				// passes the filters, one record per distinct location, 0x100 on all but candidate 0
				bool once = false;
				std::vector<std::tuple<int, unsigned long long, int>> seen;
				for (int t = 0; t < topn; ++t) {
					const View vt = view(i, t);
					if (!passes(vt)) continue;
					once = true;
					const auto key = std::make_tuple(vt.h->contig, (unsigned long long) vt.h->pos, vt.h->reverse);
					if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
					seen.push_back(key);
					write_mapped(vt, t ? 0x100 : 0, "*", 0, 0);
				}
				if (once) ++n_mapped; else write_unmapped(v, 0, -1, 0, '*', 0);
			}
		} else {
			for (int i = 0; i + 1 < n; i += 2) {
				// read1 = the first mate (even ReadId), written second by AlignmentBuffer::WriteRead; read2 = its mate
				const View v1 = view(i), v2 = view(i + 1);
				if (v1.r->seq.empty() || v2.r->seq.empty()) continue;  // GenericReadWriter.h:250-252
				n_total += 2;
				const ngm_hit &h1 = *v1.h, &h2 = *v2.h;
				// AlignmentBuffer::WriteRead (AlignmentBuffer.cpp:175-199): is the pair consistent?
				bool paired_fail = (h1.pair_flags & NGM_PAIR_FAILED) || (h2.pair_flags & NGM_PAIR_FAILED);
				if (h1.mapped && h2.mapped) {
					const long long distance = (h2.pos > h1.pos) ? (long long) (h2.pos - h1.pos) + v1.L : (long long) (h1.pos - h2.pos) + v2.L;
					if (h1.contig != h2.contig || distance < o.min_insert || distance > max_insert || h1.reverse == h2.reverse) paired_fail = true;
				}
				const bool m1 = passes(v1), m2 = passes(v2);  // GenericReadWriter::WritePair
				n_mapped += (m1 ? 1 : 0) + (m2 ? 1 : 0);
				const int f1 = 0x1 | 0x40, f2 = 0x1 | 0x80;  // SAMWriter::DoWritePair (SAMWriter.cpp:230-310)
				const unsigned long long p1 = h1.pos + 1, p2 = h2.pos + 1;
				if (!m1 && !m2) {
					write_unmapped(v2, f2 | 0x8, -1, 0, '*', 0);
					write_unmapped(v1, f1 | 0x8, -1, 0, '*', 0);
				} else if (!m1) {
					write_mapped(v2, f2 | 0x8, "=", p2, 0);
					write_unmapped(v1, f1, h2.contig, p2, '=', p2);
				} else if (!m2) {
					write_unmapped(v2, f2, h1.contig, p1, '=', p1);
					write_mapped(v1, f1 | 0x8, "=", p1, 0);
				} else if (!paired_fail) {
					if (!h1.reverse) {
						const long long d = ((long long) h2.pos + v2.L - h2.qstart - h2.qend) - (long long) h1.pos;
						write_mapped(v2, f2 | 0x2, "=", p1, -d);
						write_mapped(v1, f1 | 0x2 | 0x20, "=", p2, d);
					} else if (!h2.reverse) {
						const long long d = ((long long) h1.pos + v1.L - h1.qstart - h1.qend) - (long long) h2.pos;
						write_mapped(v2, f2 | 0x2 | 0x20, "=", p1, d);
						write_mapped(v1, f1 | 0x2, "=", p2, -d);
					}
				} else {
					write_mapped(v2, f2 | (h1.reverse ? 0x20 : 0), ngm_ref_contig_name(ref, h1.contig), p1, 0);
					write_mapped(v1, f1 | (h2.reverse ? 0x20 : 0), ngm_ref_contig_name(ref, h2.contig), p2, 0);
				}
			}
		}
		batch.clear();
	};
	auto strip_mate = [&](Read &r) {  // ReadProvider::NextRead (ReadProvider.cpp:419-422)
		const size_t L = r.name.size();
		if (L >= 2 && r.name[L - 2] == o.pe_delimiter) r.name.resize(L - 2);
	};
	if (!o.paired) {
		SeqReader in(o.qry.c_str());
		Read r;
		while (in.next(r)) {
			batch.push_back(r);
			if ((int) batch.size() == o.batch) flush();
		}
	} else {
		const bool interleaved = !o.qry.empty();  // ReadProvider::GenerateRead (ReadProvider.cpp:526-584)
		SeqReader in1(interleaved ? o.qry.c_str() : o.qry1.c_str());
		SeqReader *in2 = interleaved ? &in1 : new SeqReader(o.qry2.c_str());
		if (!in1.ok() || !in2->ok()) die("cannot open the paired-end input");
		Read a, b;
		for (;;) {
			const bool ha = in1.next(a), hb = ha ? in2->next(b) : false;
			if (!ha) break;
			if (!hb) die("Error in input file. Number of reads in input not even. Please check the input or mapped in single-end mode.");
			strip_mate(a); strip_mate(b);
			if (a.name != b.name) die("Error while reading paired end reads. Names of mates don't match: " + a.name + " and " + b.name + ".");
			batch.push_back(a); batch.push_back(b);
			if ((int) batch.size() >= (o.batch & ~1)) flush();
		}
		if (!interleaved) delete in2;
	}
	flush();
	fclose(out);
	snprintf(msg, sizeof(msg), "Done (%zu reads mapped (%.2f%%), %zu reads not mapped, %zu lines written)", n_mapped,
			n_total ? 100.0 * n_mapped / n_total : 0.0, n_total - n_mapped, n_written);
	info("MAIN", msg);
	ngm_mapper_destroy(m);
	ngm_ref_destroy(ref);
	return 0;
}
