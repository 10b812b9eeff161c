// ialignment_adapter.cpp -- NextGenMap's plugin surface over the flat C ABI.
//
// class HipAlignment is the MI355X stand-in for SWOclCigar (lib/mason/opencl/SWOclCigar.h) behind
// `class IAlignment`; the extern "C" functions are the plugin exports of
// lib/mason/opencl/SWOcl_export.cpp:20-83.  Configuration is read through IConfig with the same keys
// the OpenCL host reads (SWOcl.cpp:208-217, SWOclCigar.cpp:450-454).
#include "../../include/ngm_ialignment.h"
#include "../../include/ngm_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

ILog const *g_log = nullptr;
IConfig *g_config = nullptr;

void log_msg(int lvl, const char *msg) {
	if (g_log) g_log->_Message(lvl, "HIP", "%s", msg);
	else fprintf(stderr, "[ngm-hip] %s\n", msg);
}

class HipAlignment : public IAlignment {
public:
	HipAlignment(ngm_hip_ctx *ctx, bool alt, bool slam) : ctx_(ctx), alt_(alt), slam_(slam) {}
	~HipAlignment() override { ngm_hip_destroy(ctx_); }

	int GetScoreBatchSize() const override { return ngm_hip_score_batch_size(ctx_); }
	int GetAlignBatchSize() const override { return ngm_hip_align_batch_size(ctx_); }

	int BatchScore(int const mode, int const batchSize, char const *const *const refSeqList,
			char const *const *const qrySeqList, char const *const *const /*qalSeqList*/, float *const results,
			void *extData) override {
		if (batchSize <= 0) { log_msg(1, "Score for batchSize <= 0"); return 0; }  // SWOcl.cpp:39-42
		// bisulfite mapping: ScoreBuffer passes its direction bytes as extData (src/ScoreBuffer.cpp:93-127); the "BSMappingActive"
		// bit of include/IAlignment.h:37 is never set by the 0.5.5 callers (ScoreBuffer.h:90, AlignmentBuffer.cpp:114)
		const char *dir = (alt_ || (mode & 0x10000)) ? static_cast<const char *>(extData) : nullptr;
		int r = ngm_hip_batch_score(ctx_, mode, batchSize, refSeqList, qrySeqList, results, dir);
		if (r < 0) { log_msg(2, ngm_hip_last_error(ctx_)); return 0; }
		return r;
	}

	int BatchAlign(int const mode, int const batchSize, char const *const *const refSeqList,
			char const *const *const qrySeqList, char const *const *const /*qalSeqList*/, Align *const results,
			void *extData) override {
		if (batchSize <= 0) { log_msg(1, "Align for batchSize <= 0"); return 0; }  // SWOclCigar.cpp:109-112
		const char *dir = (alt_ || (mode & 0x10000)) ? static_cast<const char *>(extData) : nullptr;
		std::vector<ngm_hip_align_out> out(static_cast<size_t>(batchSize));
		for (int i = 0; i < batchSize; ++i) { out[i].cigar = results[i].pBuffer1; out[i].md = results[i].pBuffer2; }
		int r = ngm_hip_batch_align(ctx_, mode, batchSize, refSeqList, qrySeqList, out.data(), dir);
		if (r < 0) { log_msg(2, ngm_hip_last_error(ctx_)); return 0; }
		for (int i = 0; i < batchSize; ++i) {
			results[i].PositionOffset = out[i].position_offset;
			results[i].QStart = out[i].qstart;
			results[i].QEnd = out[i].qend;
			results[i].Score = out[i].score_token;
			results[i].Identity = out[i].identity;
			results[i].NM = out[i].nm;
			if (slam_ && out[i].score_token != -1.0f) results[i].ExtendedData = slam_records(refSeqList[i] + out[i].position_offset, qrySeqList[i], out[i]);
		}
		return r;
	}

private:
	// SLAM-seq: the per-column records computeCigarMD leaves behind Align::ExtendedData (lib/mason/opencl/SWOclCigar.cpp:442-447,
	// :484-497, :523-536) for the TC / RA / MP tags of the writers (src/writer/GenericReadWriter.h:87-187): one per '=' / 'X'
	// column -- type = 5 * class(ref) + class(read), read and alignment-relative reference position, match -- closed by the default
	// record (type -1).  The columns come from the CIGAR, '=' or 'X' from the MD string (a letter outside '^' runs is an 'X'
	// column of the device's element list).  Freed by the caller with delete[] (src/MappedRead.cpp:97-99).
	static void *slam_records(const char *ref, const char *qry, const ngm_hip_align_out &o) {
		static const auto cls = [](char ch) -> int {
			switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
		};
		size_t cols = 0;
		for (const char *c = o.cigar; *c;) { int n = 0; while (*c >= '0' && *c <= '9') n = n * 10 + (*c++ - '0'); if (!*c) break; if (*c == 'M') cols += (size_t) n; ++c; }
		AlignmentPosition *rec = new AlignmentPosition[cols + 1], *w = rec;
		const char *md = o.md;
		long eq_left = 0;
		int read_i = o.qstart, ref_i = 0;
		for (const char *c = o.cigar; *c;) {
			int n = 0;
			while (*c >= '0' && *c <= '9') n = n * 10 + (*c++ - '0');
			if (!*c) break;
			const char op = *c++;
			if (op == 'M') {
				for (int k = 0; k < n; ++k, ++w) {
					bool match = true;
					for (;;) {
						if (eq_left > 0) { --eq_left; break; }
						if (*md >= '0' && *md <= '9') { while (*md >= '0' && *md <= '9') eq_left = eq_left * 10 + (*md++ - '0'); if (eq_left == 0 && !(*md >= 'A' && *md <= 'Z') && !(*md >= 'a' && *md <= 'z')) break; continue; }
						if (*md && *md != '^') { ++md; match = false; }
						break;
					}
					w->type = 5 * cls(ref[ref_i + k]) + cls(qry[read_i + k]);
					w->readPosition = read_i + k;
					w->refPosition = ref_i + k;
					w->match = match;
				}
				read_i += n; ref_i += n;
			} else if (op == 'I') {
				read_i += n;
			} else if (op == 'D') {
				while (*md >= '0' && *md <= '9') ++md;   // the "0" (or the matches already consumed) in front of '^'
				eq_left = 0;
				if (*md == '^') { ++md; for (int k = 0; k < n && *md; ++k) ++md; }
				ref_i += n;
			}
		}
		return rec;
	}

	ngm_hip_ctx *ctx_;
	bool alt_, slam_;
};

bool integral(float v, int *out) {
	const float r = std::round(v);
	*out = static_cast<int>(r);
	return std::fabs(v - r) < 1e-6f;
}

}  // namespace

extern "C" {

void SetLog(ILog const *log) { g_log = log; }
void SetConfig(IConfig *config) { g_config = config; }
int Cookie() { return cCookie; }
bool IsAvailable() { return ngm_hip_device_count() > 0; }

// mode: low byte = device ordinal, byte 1 = report type (1 = CIGAR + MD; 0 = plain text, which NGM
// 0.5.5 no longer instantiates either: src/NGM.cpp:407-416).
IAlignment *CreateAlignment(int const mode) {
	if (!g_config) { log_msg(2, "CreateAlignment called before SetConfig"); return nullptr; }
	const int report = (mode >> 8) & 0xFF;
	if (report != 1) { log_msg(2, "Unsupported report type (only CIGAR + MD output is implemented)"); return nullptr; }
	IConfig &cfg = *g_config;
	const bool bs = cfg.Exists("bs_mapping") && cfg.GetInt("bs_mapping") == 1;
	const int slam = cfg.Exists("slam_seq") ? cfg.GetInt("slam_seq") : 0;
	ngm_hip_params p{};
	p.abi_version = NGM_HIP_ABI_VERSION;
	p.qry_max_len = cfg.GetInt("qry_max_len");
	p.corridor = cfg.GetInt("corridor");
	if (!integral(cfg.GetFloat("match_bonus"), &p.match_bonus) || !integral(cfg.GetFloat("mismatch_penalty"), &p.mismatch_penalty) ||
			!integral(cfg.GetFloat("gap_read_penalty"), &p.gap_read_penalty) || !integral(cfg.GetFloat("gap_ref_penalty"), &p.gap_ref_penalty)) {
		log_msg(2, "the HIP backend needs integer scores (match_bonus, mismatch_penalty, gap_read_penalty, gap_ref_penalty)");
		return nullptr;
	}
	p.variant = NGM_VARIANT_OCL_GPU;
	if (const char *v = getenv("NGM_HIP_VARIANT")) p.variant = atoi(v) ? NGM_VARIANT_OCL_CPU : NGM_VARIANT_OCL_GPU;
	p.hard_clip = cfg.Exists("hard_clip") ? cfg.GetInt("hard_clip") : 0;
	p.silent_clip = cfg.Exists("silent_clip") ? cfg.GetInt("silent_clip") : 0;
	p.max_batch = 0;
	// `--affine` swaps the plugin for EndToEndAffine in the reference (src/NGM.cpp:397-404); here it is a personality
	p.personality = (cfg.Exists("affine") && cfg.GetInt("affine")) ? NGM_PERSONALITY_AFFINE : NGM_PERSONALITY_LINEAR;
	if (p.personality == NGM_PERSONALITY_AFFINE && !integral(cfg.GetFloat("gap_extend_penalty"), &p.gap_extend_penalty)) {
		log_msg(2, "the HIP backend needs an integer gap_extend_penalty");
		return nullptr;
	}
	if (bs || (slam & 2)) {  // lib/mason/opencl/SWOcl.cpp:225-242 (SLAM-seq negates match_bonus_tc itself, :237: the engine does the same)
		p.alt_scoring = bs ? NGM_ALT_BISULFITE : NGM_ALT_SLAMSEQ;
		if (!integral(cfg.GetFloat("match_bonus_tt"), &p.match_bonus_tt) || !integral(cfg.GetFloat("match_bonus_tc"), &p.match_bonus_tc)) {
			log_msg(2, "the HIP backend needs integer scores (match_bonus_tt, match_bonus_tc)");
			return nullptr;
		}
	}
	// computeCigarMD's conversion rule (SWOclCigar.cpp:301-320): slam_seq, any value, wins over bs_mapping
	p.alt_cigar = slam ? NGM_ALT_SLAMSEQ : (bs ? NGM_ALT_BISULFITE : NGM_ALT_NONE);
	ngm_hip_ctx *ctx = ngm_hip_create(mode & 0xFF, &p);
	if (!ctx) { log_msg(2, ngm_hip_last_error(nullptr)); return nullptr; }
	return new HipAlignment(ctx, bs || slam != 0, slam != 0);
}

void DeleteAlignment(IAlignment *instance) { delete instance; }
void ExternalDeleteString(char *mem) { delete[] mem; }

}  // extern "C"
