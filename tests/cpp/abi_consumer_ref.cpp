// Compiled against the REFERENCE's headers (by -I path; nothing is copied): calls every virtual of an
// IAlignment / IConfig / ILog that was IMPLEMENTED in a TU built against include/ngm_ialignment.h.
// If the two header sets disagree on vtable order or struct layout, the returned codes differ.
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "IAlignment.h"
#include "IConfig.h"
#include "ILog.h"

ILog const *_log = 0;
IConfig *_config = 0;

extern "C" IAlignment *abi_make_alignment();
extern "C" IConfig *abi_make_config();
extern "C" ILog *abi_make_log();
extern "C" int abi_align_layout(int what);

int main() {
	int bad = 0;
	IAlignment *a = abi_make_alignment();
	if (a->GetScoreBatchSize() != 1001) bad |= 1;
	if (a->GetAlignBatchSize() != 1002) bad |= 2;
	float r = 0;
	if (a->BatchScore(0, 5, 0, 0, 0, &r, 0) != 1003) bad |= 4;
	Align al;
	if (a->BatchAlign(0, 6, 0, 0, 0, &al, 0) != 1004) bad |= 8;
	if (al.PositionOffset != 11 || al.QStart != 12 || al.QEnd != 13 || al.NM != 14 || al.Score != 15.0f || al.Identity != 16.0f) bad |= 16;
	delete a;
	IConfig *c = abi_make_config();
	if (strcmp(c->GetString("x"), "s") != 0) bad |= 32;
	if (c->GetInt("x") != 2001 || c->GetInt("x", 0, 1) != 2002 || c->GetParameter("x") != 2003) bad |= 64;
	if (c->GetFloat("x") != 2004.0f || c->GetFloat("x", 0.f, 1.f) != 2005.0f) bad |= 128;
	if (c->GetIntArray("x", 0, 0) != 2006 || c->GetFloatArray("x", 0, 0) != 2007 || c->GetDoubleArray("x", 0, 0) != 2008) bad |= 256;
	if (!c->Exists("x") || c->HasArray("x")) bad |= 512;
	delete c;
	ILog *l = abi_make_log();
	l->_Message(1, "t", "m");
	l->_Debug(1, "t", "m");
	if (l->null != (void *) 0x1234) bad |= 1024;
	delete l;
	if (abi_align_layout(0) != (int) sizeof(Align) || abi_align_layout(1) != (int) offsetof(Align, pBuffer2) ||
			abi_align_layout(2) != (int) offsetof(Align, PositionOffset) || abi_align_layout(3) != (int) offsetof(Align, NM) ||
			abi_align_layout(4) != cCookie)
		bad |= 2048;
	printf("abi mismatch mask: %d\n", bad);
	return bad ? 1 : 0;
}
