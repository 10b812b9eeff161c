/*
 * ngm_ialignment.h -- C++ view of the drop-in boundary: the abstract interfaces NextGenMap hands to /
 * expects from an alignment plugin, restated so that the adapter in nextgenmap_amd/csrc can be built
 * without the NextGenMap tree.  Layouts and virtual-function ORDER are ABI (Itanium vtables), they
 * must match the reference headers exactly:
 *     struct Align, class IAlignment, cCookie   include/IAlignment.h:14-72
 *     class IConfig                             include/IConfig.h:61-81
 *     class ILog + level bits                   include/ILog.h:4-30
 * When building inside the NextGenMap tree, include its own headers instead and define
 * NGM_USE_REFERENCE_HEADERS.
 *
 * Plugin exports implemented by the adapter (lib/mason/opencl/SWOcl_export.cpp:20-83; loader
 * protocol src/core/unix.cpp:169-203):
 *     void SetLog(ILog const*), void SetConfig(IConfig*), int Cookie(), bool IsAvailable(),
 *     IAlignment* CreateAlignment(int mode), void DeleteAlignment(IAlignment*),
 *     void ExternalDeleteString(char*)
 */
#ifndef NGM_IALIGNMENT_H
#define NGM_IALIGNMENT_H

#ifndef NGM_USE_REFERENCE_HEADERS

struct AlignmentPosition {
	AlignmentPosition() : type(-1), readPosition(0), refPosition(0), match(true) {}
	int type;
	int readPosition;
	int refPosition;
	bool match;
};

/* Result slot of BatchAlign; pBuffer1 / pBuffer2 are caller-allocated (4 * qry_max_len bytes). */
struct Align {
	Align() : pBuffer1(0), pBuffer2(0), ExtendedData(0), PositionOffset(0), QStart(0), QEnd(0), Score(0.0f),
			Identity(0.0f), NM(0) {}
	char *pBuffer1;      /* CIGAR */
	char *pBuffer2;      /* MD */
	void *ExtendedData;  /* SLAM-seq per-base records, unused here */
	int PositionOffset;  /* window offset where the alignment starts */
	int QStart;          /* read bases clipped at the start */
	int QEnd;            /* read bases clipped at the end */
	float Score;
	float Identity;
	int NM;
};

static int const cCookie = 0x10201130;

/* mode: bits 0-7 alignment type (0 local, 1 end-to-end), bits 8-15 report type (1 = CIGAR + MD),
 * bit 16 bisulfite mapping (extData = per-pair strand flags). */
class IAlignment {
public:
	virtual ~IAlignment() {}
	virtual int GetScoreBatchSize() const = 0;
	virtual int GetAlignBatchSize() const = 0;
	virtual int BatchScore(int const mode, int const batchSize, char const *const *const refSeqList,
			char const *const *const qrySeqList, char const *const *const qalSeqList, float *const results,
			void *extData) = 0;
	virtual int BatchAlign(int const mode, int const batchSize, char const *const *const refSeqList,
			char const *const *const qrySeqList, char const *const *const qalSeqList, Align *const results,
			void *extData) = 0;
};

class IConfig {
public:
	virtual char const *GetString(char const *const name) const = 0;
	virtual int GetInt(char const *const name) const = 0;
	virtual int GetInt(char const *const name, int min, int max) const = 0;
	virtual int GetParameter(char const *const name) const = 0;
	virtual float GetFloat(char const *const name) const = 0;
	virtual float GetFloat(char const *const name, float min, float max) const = 0;
	virtual int GetIntArray(char const *const name, int *pData, int len) const = 0;
	virtual int GetFloatArray(char const *const name, float *pData, int len) const = 0;
	virtual int GetDoubleArray(char const *const name, double *pData, int len) const = 0;
	virtual bool Exists(char const *const name) const = 0;
	virtual bool HasArray(char const *const name) const = 0;
	virtual ~IConfig() {}
};

class ILog {
public:
	virtual void _Message(int const lvl, char const *const title, char const *const msg, ...) const = 0;
	virtual void _Debug(int const lvl, char const *const title, char const *const msg, ...) const = 0;
	virtual ~ILog() {}
	void *null;
};

#endif /* NGM_USE_REFERENCE_HEADERS */

extern "C" {
void SetLog(ILog const *log);
void SetConfig(IConfig *config);
int Cookie();
bool IsAvailable();
IAlignment *CreateAlignment(int const mode);
void DeleteAlignment(IAlignment *instance);
void ExternalDeleteString(char *mem);
}

#endif
