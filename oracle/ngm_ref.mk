# oracle/ngm_ref.mk -- builds the REFERENCE program itself (ngm-core / ngm-core-debug) straight from
# its sources under $(NGM_REFERENCE) with g++; no cmake, no copies of reference sources in the repo, no
# stand-ins: its generated headers come from the reference's own generator (lib/mason/opencl/tools/oclTool.cpp),
# zlib and the OpenCL ICD loader are the system's.  Without an OpenCL device only `--affine` (SeqAn) runs
# (src/NGM.cpp:391-396) -- that is the personality this binary is the oracle for, together with everything
# above IAlignment (candidate search, score selection, MAPQ, SAM).
# Outputs: oracle/_ref/ngm/{ngm-core,ngm-core-debug}   (git-ignored)
NGM_REFERENCE ?= /root/reference
R := $(NGM_REFERENCE)
OUT := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))_ref/ngm
GEN := $(OUT)/gen
CXX ?= g++
CXXFLAGS_COMMON := -std=gnu++11 -fpermissive -w -D_BAM -pthread
INC := -I$(R)/lib/seqan-library-1.4.1/include -I$(R)/lib/bamtools-2.3.0/src -I$(R)/lib/mason/opencl -I$(R)/include \
       -I$(R)/src/parser -I$(R)/src/writer -I$(R)/src/core -I$(R)/src/misc -I$(R)/src/log -I$(R)/src/config -I$(R)/src -I$(GEN)

NGM_SRC := parser/BamParser.cpp writer/BAMWriter.cpp parser/VcfParser.cpp config/Config.cpp CS.cpp CSstatic.cpp misc/Debug.cpp \
           log/Logging.cpp MappedRead.cpp NGM_main.cpp NGM.cpp UpdateCheck.cpp core/NGMTask.cpp AlignmentBuffer.cpp PrefixTable.cpp \
           ReadProvider.cpp parser/SamParser.cpp writer/SAMWriter.cpp writer/ScoreWriter.cpp seqan/EndToEndAffine.cpp \
           SequenceProvider.cpp OutputReadBuffer.cpp ScoreBuffer.cpp core/unix.cpp core/unix_threads.cpp core/windows_threads.cpp core/windows.cpp
OCL_SRC := OclHost.cpp SWOcl.cpp SWOclAlignment.cpp SWOclCigar.cpp Timer.cpp
BAM_SRC := $(patsubst $(R)/%,%,$(shell find $(R)/lib/bamtools-2.3.0/src/api -name '*.cpp' -not -name '*_win_p.cpp' 2>/dev/null))
GEN_HDR := $(GEN)/oclDefines.h $(GEN)/oclSwScore.h $(GEN)/oclEndFreeScore.h $(GEN)/oclSwAlignment.h $(GEN)/oclSwCigar.h

REL_OBJ := $(addprefix $(OUT)/rel/src/,$(NGM_SRC:.cpp=.o)) $(addprefix $(OUT)/rel/ocl/,$(OCL_SRC:.cpp=.o)) $(addprefix $(OUT)/bam/,$(BAM_SRC:.cpp=.o))
DBG_OBJ := $(addprefix $(OUT)/dbg/src/,$(NGM_SRC:.cpp=.o)) $(addprefix $(OUT)/dbg/ocl/,$(OCL_SRC:.cpp=.o)) $(addprefix $(OUT)/bam/,$(BAM_SRC:.cpp=.o))

all: $(OUT)/ngm-core $(OUT)/ngm-core-debug

$(OUT)/oclTool: $(R)/lib/mason/opencl/tools/oclTool.cpp
	@mkdir -p $(dir $@)
	$(CXX) -w -o $@ $<
$(GEN)/%.h: $(R)/lib/mason/opencl/opencl/%.cl $(OUT)/oclTool
	@mkdir -p $(GEN)
	$(OUT)/oclTool $* $< $@

$(OUT)/rel/src/%.o: $(R)/src/%.cpp $(GEN_HDR)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O2 -DNDEBUG $(INC) -c $< -o $@
$(OUT)/dbg/src/%.o: $(R)/src/%.cpp $(GEN_HDR)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O1 -DDEBUGLOG $(INC) -c $< -o $@
$(OUT)/rel/ocl/%.o: $(R)/lib/mason/opencl/%.cpp $(GEN_HDR)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O2 -DNDEBUG $(INC) -c $< -o $@
$(OUT)/dbg/ocl/%.o: $(R)/lib/mason/opencl/%.cpp $(GEN_HDR)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O1 -DDEBUGLOG $(INC) -c $< -o $@
$(OUT)/bam/%.o: $(R)/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O2 -DNDEBUG -I$(R)/lib/bamtools-2.3.0/src -c $< -o $@

$(OUT)/ngm-core: $(REL_OBJ)
	$(CXX) -pthread -o $@ $^ -lz -lOpenCL
$(OUT)/ngm-core-debug: $(DBG_OBJ)
	$(CXX) -pthread -o $@ $^ -lz -lOpenCL

.PHONY: all

# ---- the reference's affine IAlignment (SeqAn) behind a small driver of ours --------------------------------
AFF_SRC := seqan/EndToEndAffine.cpp config/Config.cpp log/Logging.cpp core/unix.cpp core/unix_threads.cpp
AFF_OBJ := $(addprefix $(OUT)/rel/src/,$(AFF_SRC:.cpp=.o))
$(OUT)/affine_ref_main.o: $(dir $(abspath $(lastword $(MAKEFILE_LIST))))affine_ref_main.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -O2 -DNDEBUG $(INC) -I$(R)/src/seqan -c $< -o $@
$(OUT)/ngm_affine_ref: $(OUT)/affine_ref_main.o $(AFF_OBJ)
	$(CXX) -pthread -o $@ $^ -lz
affine: $(OUT)/ngm_affine_ref
all: affine
.PHONY: affine

# ---- the reference's SWOclCigar::computeCigarMD (default personality: CIGAR / MD / NM / Identity) behind a driver of ours ----
# The method is private and SWOcl's constructor needs an OpenCL device: the driver (and only the driver) is compiled with
# -fno-access-control and calls the method on zeroed storage with alignment_length set.  The objects are the ones ngm-core links.
LIN_OBJ := $(addprefix $(OUT)/rel/ocl/,$(OCL_SRC:.cpp=.o)) $(addprefix $(OUT)/rel/src/,config/Config.o log/Logging.o core/unix.o core/unix_threads.o)
$(OUT)/linear_cigar_ref_main.o: $(dir $(abspath $(lastword $(MAKEFILE_LIST))))linear_cigar_ref_main.cpp $(GEN_HDR)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -fno-access-control -O2 -DNDEBUG $(INC) -c $< -o $@
$(OUT)/ngm_linear_cigar_ref: $(OUT)/linear_cigar_ref_main.o $(LIN_OBJ)
	$(CXX) -pthread -o $@ $^ -lz -lOpenCL
linear_cigar: $(OUT)/ngm_linear_cigar_ref
all: linear_cigar
.PHONY: linear_cigar
