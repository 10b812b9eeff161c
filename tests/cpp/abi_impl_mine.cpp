// Built against include/ngm_ialignment.h: implementations whose virtuals return distinct codes.
#include <cstddef>

#include "ngm_ialignment.h"

namespace {
struct A : IAlignment {
	int GetScoreBatchSize() const override { return 1001; }
	int GetAlignBatchSize() const override { return 1002; }
	int BatchScore(int const, int const, char const *const *const, char const *const *const, char const *const *const, float *const, void *) override { return 1003; }
	int BatchAlign(int const, int const, char const *const *const, char const *const *const, char const *const *const, Align *const r, void *) override {
		r->PositionOffset = 11; r->QStart = 12; r->QEnd = 13; r->NM = 14; r->Score = 15.0f; r->Identity = 16.0f; return 1004; }
};
struct Cfg : IConfig {
	char const *GetString(char const *const) const override { return "s"; }
	int GetInt(char const *const) const override { return 2001; }
	int GetInt(char const *const, int, int) const override { return 2002; }
	int GetParameter(char const *const) const override { return 2003; }
	float GetFloat(char const *const) const override { return 2004.0f; }
	float GetFloat(char const *const, float, float) const override { return 2005.0f; }
	int GetIntArray(char const *const, int *, int) const override { return 2006; }
	int GetFloatArray(char const *const, float *, int) const override { return 2007; }
	int GetDoubleArray(char const *const, double *, int) const override { return 2008; }
	bool Exists(char const *const) const override { return true; }
	bool HasArray(char const *const) const override { return false; }
};
struct L : ILog {
	L() { null = (void *) 0x1234; }
	void _Message(int const, char const *const, char const *const, ...) const override {}
	void _Debug(int const, char const *const, char const *const, ...) const override {}
};
}  // namespace

extern "C" IAlignment *abi_make_alignment() { return new A(); }
extern "C" IConfig *abi_make_config() { return new Cfg(); }
extern "C" ILog *abi_make_log() { return new L(); }
extern "C" int abi_align_layout(int what) {
	switch (what) {
	case 0: return (int) sizeof(Align);
	case 1: return (int) offsetof(Align, pBuffer2);
	case 2: return (int) offsetof(Align, PositionOffset);
	case 3: return (int) offsetof(Align, NM);
	default: return cCookie;
	}
}
// the plugin exports are declared by the header; give the linker bodies for this test binary
extern "C" { void SetLog(ILog const *) {} void SetConfig(IConfig *) {} int Cookie() { return cCookie; } bool IsAvailable() { return false; }
IAlignment *CreateAlignment(int const) { return 0; } void DeleteAlignment(IAlignment *) {} void ExternalDeleteString(char *) {} }
