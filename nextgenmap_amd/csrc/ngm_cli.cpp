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
	std::string ref, qry, out;
	int device = 0, kmer = 13, kmer_skip = 2, bin_size = 2, mode = 0, corridor = -1, max_read_length = 0, min_mq = 0, max_kfreq = 0;
	int match = 10, mismatch = 15, gap_read = -1, gap_ref = -1, gap_extend = -1, affine = 0, hard_clip = 0, silent_clip = 0, no_unal = 0, max_cmrs = 2147483647;
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
		SKIPSAVE, BATCH, VARIANT, AFFINE, GEP, UNSUPPORTED };
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
		{"qry1", required_argument, 0, UNSUPPORTED}, {"qry2", required_argument, 0, UNSUPPORTED}, {"paired", no_argument, 0, UNSUPPORTED},
		{"affine", no_argument, 0, AFFINE}, {"gap-extend-penalty", required_argument, 0, GEP}, {"bam", no_argument, 0, UNSUPPORTED}, {"bs-mapping", no_argument, 0, UNSUPPORTED},
		{"slam-seq", required_argument, 0, UNSUPPORTED}, {"topn", required_argument, 0, UNSUPPORTED}, {"strata", no_argument, 0, UNSUPPORTED},
		{"argos", no_argument, 0, UNSUPPORTED}, {"vcf", required_argument, 0, UNSUPPORTED}, {"config", required_argument, 0, UNSUPPORTED},
		{0, 0, 0, 0}};
	int c, idx = 0;
	while ((c = getopt_long(argc, argv, "o:q:r:t:gs:k:lei:R:C:Q:", lo, &idx)) != -1) {
		switch (c) {
		case 'r': o.ref = optarg; break;
		case 'q': o.qry = optarg; break;
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
		case GEP: o.gap_extend = atoi(optarg); break;
		case MAXCMRS: o.max_cmrs = atoi(optarg); break;
		case NOUNAL: o.no_unal = 1; break;
		case NOPROG: case SKIPSAVE: break;
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
	ngm_ref *ref = ngm_ref_create_from_fasta(o.device, &rp, o.ref.c_str());
	if (!ref) die(ngm_pipeline_last_error());
	info("PREPROCESS", "index entries: " + std::to_string(ngm_ref_index_entries(ref)) + ", max. k-mer frequency " +
			std::to_string(o.max_kfreq > 0 ? o.max_kfreq : ngm_ref_auto_max_kfreq(ref)));
	if (o.qry.empty()) { ngm_ref_destroy(ref); return 0; }  // index only, like `ngm -r ref.fa`
	if (o.out.empty()) die("no output file given (-o/--output)");

	// ---- pass 1: read lengths + the sample for the sensitivity estimate (ReadProvider.cpp:201-305) ----------
	size_t max_len = 0, min_len = 9999999, sum_len = 0, count = 0;
	std::vector<Read> sample;
	{
		SeqReader in(o.qry.c_str());
		if (!in.ok()) die("cannot open " + o.qry);
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
		if (n_used > 0) {
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

	// ---- pass 2: map in batches ------------------------------------------------------------------------------
	SeqReader in(o.qry.c_str());
	std::vector<Read> batch;
	std::vector<char> rows, cig, md;
	std::vector<ngm_hit> hits;
	size_t n_total = 0, n_mapped = 0, n_written = 0;
	const size_t stride = (size_t) 4 * q;
	auto flush = [&]() {
		const int n = (int) batch.size();
		if (n == 0) return;
		rows.assign((size_t) n * q, 0);
		for (int i = 0; i < n; ++i) pack_row(batch[i], q, &rows[(size_t) i * q]);
		hits.resize(n); cig.resize((size_t) n * stride); md.resize((size_t) n * stride);
		if (ngm_mapper_map_se(m, n, rows.data(), hits.data(), cig.data(), md.data()) < 0) die(ngm_pipeline_last_error());
		std::string rev;
		for (int i = 0; i < n; ++i) {
			const Read &r = batch[i];
			const ngm_hit &h = hits[i];
			const char *row = &rows[(size_t) i * q];
			const int L = (int) strnlen(row, q);
			if (r.seq.empty()) continue;  // NGMNames::Empty reads are discarded (GenericReadWriter.h:245-247)
			const char *qual = r.qual.empty() ? "*" : r.qual.c_str();
			const int qual_len = r.qual.empty() ? 1 : std::min<int>((int) r.qual.size(), L);
			float min_res = o.min_residues;
			if (min_res <= 1.0f) min_res = L * min_res;
			bool mapped = h.mapped && h.mapq >= o.min_mq && h.identity >= o.min_identity && (float) (L - h.qstart - h.qend) >= min_res;
			++n_total;
			if (!mapped) {
				if (o.no_unal) continue;
				fprintf(out, "%s\t4\t*\t0\t0\t*\t*\t0\t0\t%.*s\t%.*s\n", r.name.c_str(), L, row, qual_len, qual);
				++n_written;
				continue;
			}
			++n_mapped; ++n_written;
			int flags = 0;
			std::string seq(row, L), ql(qual, qual_len);
			if (h.reverse) {
				flags |= 0x10;
				for (int t = 0; t < L; ++t) { const char ch = row[L - 1 - t]; seq[t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch; }
				if (ql[0] != '*' || ql.size() > 1) std::reverse(ql.begin(), ql.end());
			}
			const bool clip = o.hard_clip || o.silent_clip;
			const int s0 = clip ? h.qstart : 0, sl = clip ? L - h.qstart - h.qend : L;
			const float identity = roundf(h.identity * 10000.0f) / 10000.0f;
			fprintf(out, "%s\t%d\t%s\t%llu\t%d\t%s\t*\t0\t0\t%.*s\t%.*s\tAS:i:%d\tNM:i:%d\tNH:i:%d\tXI:f:%g\tX0:i:%d\tXE:i:%d\tXR:i:%d\tMD:Z:%s\n",
					r.name.c_str(), flags, ngm_ref_contig_name(ref, h.contig), (unsigned long long) h.pos + 1, h.mapq, &cig[(size_t) i * stride],
					sl, seq.c_str() + s0, (ql.size() == 1 && ql[0] == '*') ? 1 : sl, (ql.size() == 1 && ql[0] == '*') ? "*" : ql.c_str() + s0,
					(int) h.score, h.nm, h.n_best, identity, h.n_best, (int) h.max_votes, L - h.qstart - h.qend, &md[(size_t) i * stride]);
		}
		batch.clear();
	};
	Read r;
	while (in.next(r)) {
		batch.push_back(r);
		if ((int) batch.size() == o.batch) flush();
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
