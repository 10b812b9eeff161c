// ngm_host_driver.cpp -- drives the plugin the way NextGenMap's host does:
//   SetLog -> SetConfig -> Cookie check -> CreateAlignment(gpu | 1<<8)      (src/core/unix.cpp:169-203, src/CS.cpp:456)
//   ScoreBuffer::DoRun : results preset to -1, BatchScore(mode, n, ref, qry, 0, scores, 0)   (src/ScoreBuffer.cpp:113-132)
//   AlignmentBuffer::DoRun : pBuffer1/2 = new char[4*qry_max_len] stamped "!!!", BatchAlign(mode | 1<<8, ...) (src/AlignmentBuffer.cpp:100-120)
// usage: ngm_host_driver <in.bin> <out.txt> <mode>
//   in.bin : int32 n, q, c; n*(q+c) ref bytes; n*q qry bytes
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ngm_ialignment.h"

struct TestConfig : public IConfig {
	std::map<std::string, std::string> kv;
	char const *GetString(char const *const name) const override { auto it = kv.find(name); return it == kv.end() ? "" : it->second.c_str(); }
	int GetInt(char const *const name) const override { return atoi(GetString(name)); }
	int GetInt(char const *const name, int, int) const override { return GetInt(name); }
	int GetParameter(char const *const name) const override { return GetInt(name); }
	float GetFloat(char const *const name) const override { return (float) atof(GetString(name)); }
	float GetFloat(char const *const name, float, float) const override { return GetFloat(name); }
	int GetIntArray(char const *const, int *, int) const override { return 0; }
	int GetFloatArray(char const *const, float *, int) const override { return 0; }
	int GetDoubleArray(char const *const, double *, int) const override { return 0; }
	bool Exists(char const *const name) const override { return kv.count(name) != 0; }
	bool HasArray(char const *const) const override { return false; }
};

struct TestLog : public ILog {
	void _Message(int const lvl, char const *const, char const *const msg, ...) const override {
		va_list ap; va_start(ap, msg); fprintf(stderr, "[log %d] ", lvl); vfprintf(stderr, msg, ap); fputc('\n', stderr); va_end(ap);
	}
	void _Debug(int const, char const *const, char const *const, ...) const override {}
};

int main(int argc, char **argv) {
	if (argc < 4) return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	int hdr[3];
	if (fread(hdr, 4, 3, f) != 3) return 2;
	const int n = hdr[0], q = hdr[1], c = hdr[2], mode = atoi(argv[3]);
	std::vector<char> ref((size_t) n * (q + c + 2)), qry((size_t) n * q);
	// window rows are refMaxLen = ((q+c)|1)+1 bytes in the host, of which q+c are read (ScoreBuffer.h:112)
	for (int i = 0; i < n; ++i) if (fread(&ref[(size_t) i * (q + c + 2)], 1, q + c, f) != (size_t) (q + c)) return 2;
	if (fread(qry.data(), 1, qry.size(), f) != qry.size()) return 2;
	fclose(f);

	TestLog log;
	TestConfig cfg;
	cfg.kv = {{"qry_max_len", std::to_string(q)}, {"corridor", std::to_string(c)}, {"match_bonus", "10"},
			{"mismatch_penalty", "15"}, {"gap_read_penalty", "20"}, {"gap_ref_penalty", "20"}, {"bs_mapping", "0"}};
	if (argc > 4 && std::string(argv[4]) == "affine") {  // `ngm --affine`: Config.cpp:433-439
		cfg.kv["affine"] = "1"; cfg.kv["gap_read_penalty"] = "33"; cfg.kv["gap_ref_penalty"] = "33"; cfg.kv["gap_extend_penalty"] = "3";
	}
	SetLog(&log);
	SetConfig(&cfg);
	if (Cookie() != cCookie) { fprintf(stderr, "cookie mismatch\n"); return 3; }
	if (!IsAvailable()) { fprintf(stderr, "no device\n"); return 4; }
	IAlignment *aligner = CreateAlignment(0 | (1 << 8));
	if (!aligner) return 5;
	if (aligner->GetScoreBatchSize() < n || aligner->GetAlignBatchSize() / 2 < 1) return 6;

	std::vector<const char *> refs(n), qrys(n);
	for (int i = 0; i < n; ++i) { refs[i] = &ref[(size_t) i * (q + c + 2)]; qrys[i] = &qry[(size_t) i * q]; }
	std::vector<float> scores(n, -1.0f);
	if (aligner->BatchScore(mode, n, refs.data(), qrys.data(), 0, scores.data(), 0) != n) return 7;

	std::vector<Align> al(n);
	for (int i = 0; i < n; ++i) {
		al[i].pBuffer1 = new char[std::max(1, q) * 4];
		al[i].pBuffer2 = new char[std::max(1, q) * 4];
		strcpy(al[i].pBuffer1, "!!!");
		strcpy(al[i].pBuffer2, "!!!");
	}
	if (aligner->BatchAlign(mode | (1 << 8), n, refs.data(), qrys.data(), 0, al.data(), 0) != n) return 8;

	FILE *o = fopen(argv[2], "w");
	for (int i = 0; i < n; ++i)
		fprintf(o, "%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%.9g\t%d\n", i, (int) scores[i], al[i].pBuffer1, al[i].pBuffer2,
				al[i].PositionOffset, al[i].QStart, al[i].QEnd, al[i].NM, al[i].Identity, (int) al[i].Score);
	fclose(o);
	DeleteAlignment(aligner);
	return 0;
}
