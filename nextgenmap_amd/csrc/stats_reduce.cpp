// stats_reduce.cpp -- the ONE collective of the mapping path (include/ngm_pipeline.h ngm_stats_*): the int64[8] vector
// {reads, mapped, unmapped, written, pairs_total, pairs_broken, insert_sum, insert_cnt} of every shard process (`ngm-hip -g a,b,...
// --shard-output`: one process per GPU, nothing shared on the data path) summed with ONE ncclAllReduce -- RCCL over xGMI.  What it
// replaces in the reference: the process-wide counters every CS thread of the one NextGenMap process adds to (src/NGM.cpp:172-200:
// AddMappedRead / AddUnmappedRead / AddWrittenRead / AddReadRead) and AlignmentBuffer's pair counters (src/AlignmentBuffer.cpp:175-199).
//
// librccl is loaded at run time (dlopen): a single-GPU run never touches it, and a host without it still maps -- the shard processes'
// parent then sums the vectors it receives over their pipes (ngm_cli.cpp run_sharded), which it does in any case.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include <hip/hip_runtime.h>

// The handful of RCCL declarations this file needs, stated here instead of #include <rccl/rccl.h>: the library is found at RUN time,
// and a ROCm install without the RCCL development headers must still build libngm_hip.so.  Values as in rccl.h (NCCL 2.x ABI).
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclInt64 = 4 } ncclDataType_t;
}

#include "refindex.h"   // ngm::pipeline_set_error

struct ngm_stats_comm {
	int device = 0, rank = 0, world = 1;
	ncclComm_t comm = nullptr;
	hipStream_t st = nullptr;
	int64_t *d_buf = nullptr;
};

namespace {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
};

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (r.lib) break;
		}
		if (!r.lib) return;
		r.GetUniqueId = (decltype(r.GetUniqueId)) dlsym(r.lib, "ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank)) dlsym(r.lib, "ncclCommInitRank");
		r.AllReduce = (decltype(r.AllReduce)) dlsym(r.lib, "ncclAllReduce");
		r.CommDestroy = (decltype(r.CommDestroy)) dlsym(r.lib, "ncclCommDestroy");
		r.GetErrorString = (decltype(r.GetErrorString)) dlsym(r.lib, "ncclGetErrorString");
		r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
	});
	return r;
}

const char *nccl_err(ncclResult_t e) { return rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"; }

}  // namespace

extern "C" {

int ngm_stats_unique_id(char hex[2 * NCCL_UNIQUE_ID_BYTES + 1]) {
	Rccl &R = rccl();
	if (!R.ok) { ngm::pipeline_set_error("librccl could not be loaded"); return -38; }
	ncclUniqueId id;
	const ncclResult_t e = R.GetUniqueId(&id);
	if (e != ncclSuccess) { ngm::pipeline_set_error("ncclGetUniqueId: %s", nccl_err(e)); return -5; }
	static const char digits[] = "0123456789abcdef";
	for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) { hex[2 * i] = digits[(unsigned char) id.internal[i] >> 4]; hex[2 * i + 1] = digits[(unsigned char) id.internal[i] & 15]; }
	hex[2 * NCCL_UNIQUE_ID_BYTES] = 0;
	return 0;
}

ngm_stats_comm *ngm_stats_comm_create(int device, int rank, int world, const char *hex) {
	Rccl &R = rccl();
	if (!R.ok) { ngm::pipeline_set_error("librccl could not be loaded"); return nullptr; }
	if (!hex || strlen(hex) != 2 * NCCL_UNIQUE_ID_BYTES || rank < 0 || rank >= world) { ngm::pipeline_set_error("ngm_stats_comm_create: bad unique id / rank"); return nullptr; }
	ncclUniqueId id;
	for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) {
		auto nib = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0; };
		id.internal[i] = (char) ((nib(hex[2 * i]) << 4) | nib(hex[2 * i + 1]));
	}
	if (hipSetDevice(device) != hipSuccess) { ngm::pipeline_set_error("hipSetDevice(%d) failed", device); return nullptr; }
	ngm_stats_comm *c = new ngm_stats_comm();
	c->device = device; c->rank = rank; c->world = world;
	const ncclResult_t e = R.CommInitRank(&c->comm, world, id, rank);
	if (e != ncclSuccess) { ngm::pipeline_set_error("ncclCommInitRank (rank %d of %d, device %d): %s", rank, world, device, nccl_err(e)); delete c; return nullptr; }
	if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->d_buf, 8 * sizeof(int64_t)) != hipSuccess) {
		ngm::pipeline_set_error("out of device memory (stats all-reduce)");
		ngm_stats_comm_destroy(c);
		return nullptr;
	}
	return c;
}

int ngm_stats_allreduce(ngm_stats_comm *c, int64_t v[8]) {
	if (!c || !c->comm || !v) return -22;
	Rccl &R = rccl();
	if (hipSetDevice(c->device) != hipSuccess) return -5;
	if (hipMemcpyAsync(c->d_buf, v, 8 * sizeof(int64_t), hipMemcpyHostToDevice, c->st) != hipSuccess) { ngm::pipeline_set_error("stats all-reduce: upload failed"); return -5; }
	const ncclResult_t e = R.AllReduce(c->d_buf, c->d_buf, 8, ncclInt64, ncclSum, c->comm, c->st);
	if (e != ncclSuccess) { ngm::pipeline_set_error("ncclAllReduce: %s", nccl_err(e)); return -5; }
	if (hipMemcpyAsync(v, c->d_buf, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st) != hipSuccess || hipStreamSynchronize(c->st) != hipSuccess) {
		ngm::pipeline_set_error("stats all-reduce: download failed");
		return -5;
	}
	return 0;
}

void ngm_stats_comm_destroy(ngm_stats_comm *c) {
	if (!c) return;
	(void) hipSetDevice(c->device);
	if (c->st) { (void) hipStreamSynchronize(c->st); (void) hipStreamDestroy(c->st); }
	if (c->d_buf) (void) hipFree(c->d_buf);
	if (c->comm && rccl().CommDestroy) (void) rccl().CommDestroy(c->comm);
	delete c;
}

}  // extern "C"
