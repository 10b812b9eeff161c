// affine_ref_main.cpp -- TEST INFRASTRUCTURE: driver around the REFERENCE's own affine IAlignment
// (src/seqan/EndToEndAffine.{h,cpp} + vendored SeqAn 1.4.1, compiled from /root/reference by oracle/ngm_ref.mk).
// Only this main() is ours: it builds the reference's _Config / _Log objects, pins corridor / mode / scores via
// _Config::Override (the call ReadProvider itself uses, src/ReadProvider.cpp:296-305), feeds pairs from a binary
// file to EndToEndAffine::BatchScore / BatchAlign and dumps the results.
//   usage: ngm_affine_ref <in.bin> <out.txt> <mode> [match mismatch gap_open gap_extend]
//   in.bin: int32 n, q, c; n * (q+c) window bytes (NUL terminated strings inside the rows), n * q read bytes
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Types.h"
#include "Config.h"
#include "Log.h"
#include "EndToEndAffine.h"

#undef module_name
#define module_name "AFFREF"

// The reference defines these three in the translation unit that also holds its main() (src/NGM_main.cpp:82-83,
// :344-352), which cannot be linked next to this driver's main(): same declarations, FileSize is never called here.
ILog const *_log = 0;
IConfig *_config = 0;
uloc const FileSize(char const *const filename) {
	FILE *fp = fopen(filename, "rb");
	if (!fp) return 0;
	fseek(fp, 0, SEEK_END);
	uloc const end = ftell(fp);
	fclose(fp);
	return end;
}

int main(int argc, char **argv) {
	if (argc < 4) return 2;
	// _Config::Override cannot replace a key that already exists (std::map::insert, src/config/Config.cpp:86), so
	// everything that has a default goes through the reference's own command-line parser
	const int mode_arg = atoi(argv[3]);
	std::vector<std::string> args = {"ngm", "--affine"};
	if (mode_arg == 1) args.push_back("-e");
	if (argc >= 8) {
		args.insert(args.end(), {"--match-bonus", argv[4], "--mismatch-penalty", argv[5], "--gap-read-penalty", argv[6],
				"--gap-extend-penalty", argv[7]});
	}
	std::vector<char *> cfg_argv;
	for (auto &a : args) cfg_argv.push_back(&a[0]);
	cfg_argv.push_back(0);
	_config = new _Config((int) args.size(), cfg_argv.data());
	_log = &Log;  // static logger instance of the reference (src/log/Logging.cpp)
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	int hdr[3];
	if (fread(hdr, 4, 3, f) != 3) return 2;
	const int n = hdr[0], q = hdr[1], c = hdr[2], mode = mode_arg;
	_Config *cfg = (_Config *) _config;
	cfg->Override("corridor", c);       // keys without a default: set the way ReadProvider sets them
	cfg->Override("qry_max_len", q);
	if (cfg->GetInt(MODE, 0, 1) != mode) { fprintf(stderr, "mode not applied\n"); return 3; }
	std::vector<char> ref((size_t) n * (q + c + 1), 0), qry((size_t) n * (q + 1), 0);
	for (int i = 0; i < n; ++i) if (fread(&ref[(size_t) i * (q + c + 1)], 1, q + c, f) != (size_t) (q + c)) return 2;
	for (int i = 0; i < n; ++i) if (fread(&qry[(size_t) i * (q + 1)], 1, q, f) != (size_t) q) return 2;
	fclose(f);
	std::vector<const char *> refs(n), qrys(n);
	for (int i = 0; i < n; ++i) { refs[i] = &ref[(size_t) i * (q + c + 1)]; qrys[i] = &qry[(size_t) i * (q + 1)]; }
	EndToEndAffine aligner;
	std::vector<float> scores(n, -1.f);
	std::vector<Align> al(n);
	FILE *o = fopen(argv[2], "w");
	// one pair at a time, skipping empty sequences (SeqAn asserts length >= 1; NGM never submits them)
	for (int i = 0; i < n; ++i) {
		al[i].pBuffer1 = new char[4 * q + 16];
		al[i].pBuffer2 = new char[4 * q + 16];
		strcpy(al[i].pBuffer1, "!!!");
		strcpy(al[i].pBuffer2, "!!!");
		if (refs[i][0] == 0 || qrys[i][0] == 0) { fprintf(o, "%d\tEMPTY\n", i); continue; }
		aligner.BatchScore(mode, 1, &refs[i], &qrys[i], 0, &scores[i], 0);
		aligner.BatchAlign(mode, 1, &refs[i], &qrys[i], 0, &al[i], 0);
		fprintf(o, "%d\t%d\t%s\t%d\t%d\t%d\t%d\t%.9g\n", i, (int) scores[i], al[i].pBuffer1, al[i].PositionOffset, al[i].QStart, al[i].QEnd,
				al[i].NM, al[i].Identity);
	}
	fclose(o);
	return 0;
}
