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
	HipAlignment(ngm_hip_ctx *ctx, bool alt) : ctx_(ctx), alt_(alt) {}
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
		}
		return r;
	}

private:
	ngm_hip_ctx *ctx_;
	bool alt_;
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
	// SLAM-seq needs the per-base records behind Align::ExtendedData (SWOclCigar.cpp:442-447, :484-540) for the TC / RA / MP
	// tags (src/writer/GenericReadWriter.h:87-180): not produced here yet, so that mode is refused rather than written without them
	if (cfg.Exists("slam_seq") && cfg.GetInt("slam_seq") != 0) {
		log_msg(2, "SLAM-seq (--slam-seq) is not implemented in the HIP backend");
		return nullptr;
	}
	const bool bs = cfg.Exists("bs_mapping") && cfg.GetInt("bs_mapping") == 1;
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
	if (bs) {  // lib/mason/opencl/SWOcl.cpp:228-232
		p.alt_scoring = NGM_ALT_BISULFITE;
		if (!integral(cfg.GetFloat("match_bonus_tt"), &p.match_bonus_tt) || !integral(cfg.GetFloat("match_bonus_tc"), &p.match_bonus_tc)) {
			log_msg(2, "the HIP backend needs integer scores (match_bonus_tt, match_bonus_tc)");
			return nullptr;
		}
	}
	ngm_hip_ctx *ctx = ngm_hip_create(mode & 0xFF, &p);
	if (!ctx) { log_msg(2, ngm_hip_last_error(nullptr)); return nullptr; }
	return new HipAlignment(ctx, bs);
}

void DeleteAlignment(IAlignment *instance) { delete instance; }
void ExternalDeleteString(char *mem) { delete[] mem; }

}  // extern "C"
