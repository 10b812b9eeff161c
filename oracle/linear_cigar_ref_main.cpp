// linear_cigar_ref_main.cpp -- TEST INFRASTRUCTURE: driver around the REFERENCE's own SWOclCigar::computeCigarMD
// (lib/mason/opencl/SWOclCigar.cpp:430-615, compiled from /root/reference by oracle/ngm_ref.mk -- the same objects
// that are linked into oracle/_ref/ngm/ngm-core).  computeCigarMD is the host step of NGM's default (linear-gap)
// personality that turns the backtracking kernel's run-length output into CIGAR / MD / NM / Identity / QStart / QEnd;
// it touches no OpenCL state, only `alignment_length`, `slamSeq` and three Config keys, so it runs here on the golden
// RLE rows that the reference's own kernels produced on the MI355X (tests/golden/ngm_ocl_*.npz).
// Only this main() is ours.  The method is private: this translation unit (and only this one) is compiled with
// -fno-access-control; the object is raw zeroed storage with the two data members set, because SWOcl's constructor
// needs an OpenCL device (lib/mason/opencl/SWOcl.cpp:163-200).
//   usage: ngm_linear_cigar_ref <in.bin> <out.txt> <clip> [alt]   clip: 0 soft (default), 1 --hard-clip, 2 --silent-clip
//                                                                alt: 1 = --bs-mapping, 2 = --slam-seq (bsFrom / bsTo per pair as
//                                                                SWOclCigar::BatchAlign derives them from extData, SWOclCigar.cpp:300-317)
//   in.bin: int32 n, q, c; then per pair: (q+c) window bytes, q read bytes, 4 int16 results, (2q+c+1) int16 RLE [, alt: 1 direction byte]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "Types.h"
#include "Config.h"
#include "Log.h"
#include "SWOclCigar.h"

#undef module_name
#define module_name "LINREF"

// The reference defines these in the translation unit of its main() (src/NGM_main.cpp:82-83, :344-352), which cannot be
// linked next to this driver's main(): same declarations; FileSize is never called here.
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
	const int clip = atoi(argv[3]);
	const int alt = argc > 4 ? atoi(argv[4]) : 0;
	std::vector<std::string> args = {"ngm", "-t", "1"};  // (a bare command line makes the parser print the help and throw)
	if (alt == 1) args.push_back("--bs-mapping");   // computeCigarMD reads Config "bs_mapping" (SWOclCigar.cpp:435)
	if (clip == 1) args.push_back("--hard-clip");
	if (clip == 2) args.push_back("--silent-clip");
	std::vector<char *> cfg_argv;
	for (auto &a : args) cfg_argv.push_back(&a[0]);
	cfg_argv.push_back(0);
	_config = new _Config((int) args.size(), cfg_argv.data());
	_log = &Log;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	int hdr[3];
	if (fread(hdr, 4, 3, f) != 3) return 2;
	const int n = hdr[0], q = hdr[1], c = hdr[2];
	_Config *cfg = (_Config *) _config;
	cfg->Override("corridor", c);
	cfg->Override("qry_max_len", q);
	const int al = 2 * q + c + 1;  // SWOcl.cpp:342
	// raw storage instead of a constructed object (no OpenCL device here); the method reads these two members only
	void *raw = calloc(1, sizeof(SWOclCigar));
	SWOclCigar *obj = reinterpret_cast<SWOclCigar *>(raw);
	obj->alignment_length = al;
	*const_cast<int *>(&obj->slamSeq) = alt == 2 ? 2 : 0;   // (SWOcl.cpp:165-166: Config SLAM_SEQ)
	FILE *o = fopen(argv[2], "w");
	if (!o) return 2;
	std::vector<char> ref(q + c + 1), qry(q + 1), cigar(4 * (q + c) + 64), md(4 * (q + c) + 64);
	std::vector<short> rle(al);
	short res[4];
	for (int i = 0; i < n; ++i) {
		memset(ref.data(), 0, ref.size()); memset(qry.data(), 0, qry.size());
		if (fread(ref.data(), 1, q + c, f) != (size_t) (q + c) || fread(qry.data(), 1, q, f) != (size_t) q || fread(res, 2, 4, f) != 4 ||
				fread(rle.data(), 2, al, f) != (size_t) al) return 2;
		char dir = 0, bs_from = '0', bs_to = '0';
		if (alt) {
			if (fread(&dir, 1, 1, f) != 1) return 2;
			if (alt == 1) { if (dir == 1) { bs_from = 'A'; bs_to = 'G'; } else { bs_from = 'T'; bs_to = 'C'; } }   // SWOclCigar.cpp:302-310
			if (alt == 2) { if (dir == 1) { bs_from = 'G'; bs_to = 'A'; } else { bs_from = 'C'; bs_to = 'T'; } }   // :311-318
		}
		Align a;
		a.pBuffer1 = cigar.data(); a.pBuffer2 = md.data();
		strcpy(a.pBuffer1, "!!!"); strcpy(a.pBuffer2, "!!!");
		// the call of SWOclCigar::BatchAlign (SWOclCigar.cpp:322-328) with alignments_per_thread = 1:
		//   offset = results[3], refSeq = window + results[0], PositionOffset = results[0]
		const bool ok = obj->computeCigarMD(a, res[3], rle.data(), ref.data() + res[0], qry.data(), 0, bs_from, bs_to);
		a.PositionOffset = res[0];
		fprintf(o, "%d\t%d\t%s\t%s\t%d\t%.9g\t%d\t%d\t%d\t%.9g\n", i, ok ? 1 : 0, a.pBuffer1, a.pBuffer2, a.NM, a.Identity, a.QStart, a.QEnd,
				a.PositionOffset, ok ? a.Score : -1.0f);
	}
	fclose(o);
	fclose(f);
	return 0;
}
