#!/usr/bin/env python3
"""Applies INTEGRATION.md section A to a SCRATCH COPY of the reference's src/NGM.cpp: _NGM::CreateAlignment /
_NGM::DeleteAlignment (src/NGM.cpp:388-437) hand the job to the plugin exports of libngm_hip.so (the seven C symbols of
lib/mason/opencl/SWOcl_export.cpp:20-83) instead of constructing EndToEndAffine / OclHost + SWOclCigar.
usage: dropin_patch.py <scratch copy of NGM.cpp>      (edits it in place; nothing of the reference is committed)"""
import sys

NEW = r'''
// ---- replaced by oracle/dropin_patch.py (INTEGRATION.md section A) ---------------------------------------------------
extern "C" { void SetLog(ILog const*); void SetConfig(IConfig*); int Cookie(); bool IsAvailable();
             IAlignment* CreateAlignment(int const mode); void DeleteAlignment(IAlignment*); }

IAlignment * _NGM::CreateAlignment(int const mode) {
	static bool once = (SetLog(&Log), SetConfig(&Config), true); (void) once;
	if (Cookie() != cCookie || !IsAvailable()) { Log.Error("HIP alignment backend unavailable"); Fatal(); }
	IAlignment * instance = ::CreateAlignment(mode);  // low byte = GPU ordinal, byte 1 = report type, as before
	if (instance == 0) { Log.Error("HIP alignment backend could not be created"); Fatal(); }
	return instance;
}

void _NGM::DeleteAlignment(IAlignment* instance) {
	Log.Verbose("Delete alignment called");
	if (instance != 0) ::DeleteAlignment(instance);
}
// ---- end of the replacement -----------------------------------------------------------------------------------------

'''


def main():
    path = sys.argv[1]
    s = open(path).read()
    a = s.index("IAlignment * _NGM::CreateAlignment(int const mode) {")
    b = s.index("void _NGM::MainLoop() {")
    assert "void _NGM::DeleteAlignment(IAlignment* instance) {" in s[a:b]
    open(path, "w").write(s[:a] + NEW.lstrip("\n") + s[b:])


if __name__ == "__main__":
    main()
