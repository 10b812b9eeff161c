// gz_inflate_check.cpp -- test driver: inflate a .gz file with nextgenmap_amd/csrc/gz_inflate.h and write the text.
//   gz_inflate_check <in.gz> <out>      exit 0: text written; exit 3: the decoder refused the file (the CLI then uses zlib's reader)
#include <cstdio>

#include "../../nextgenmap_amd/csrc/gz_inflate.h"

int main(int argc, char **argv) {
	if (argc < 3) return 2;
	char *text = nullptr;
	size_t len = 0, reserved = 0;
	if (!ngm::gz::inflate_file(argv[1], &text, &len, &reserved, (size_t) 8 << 30)) return 3;
	FILE *f = fopen(argv[2], "wb");
	if (!f) return 2;
	const bool ok = fwrite(text, 1, len, f) == len;
	fclose(f);
	munmap(text, reserved);
	return ok ? 0 : 2;
}
