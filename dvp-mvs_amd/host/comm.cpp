// comm.cpp — RankComm on RCCL (ncclBroadcast / ncclAllReduce over xGMI), see comm.h.
#define __HIP_PLATFORM_AMD__ 1
#include "comm.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <thread>

namespace {
[[noreturn]] void die(const char* what, const char* detail) {
	fprintf(stderr, "RankComm: %s: %s\n", what, detail);
	exit(EXIT_FAILURE);
}
void hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) die(what, hipGetErrorString(e)); }
void nccl_ok(ncclResult_t r, const char* what) { if (r != ncclSuccess) die(what, ncclGetErrorString(r)); }
}  // namespace

RankComm::RankComm(int rank, int world, int device, const std::string& rendezvous_file, const std::string& nonce, int timeout_s)
	: rank_(rank), world_(world), device_(device) {
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	if (world_ <= 1) return;
	ncclUniqueId id;
	const std::string tag = "dvp-rccl-id " + nonce + "\n";
	if (rank_ == 0) {
		nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
		const std::string tmp = rendezvous_file + ".tmp";
		{
			std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
			f.write(tag.data(), (std::streamsize)tag.size());
			f.write(reinterpret_cast<const char*>(&id), sizeof(id));
			if (!f.good()) die("cannot write", tmp.c_str());
		}
		std::filesystem::rename(tmp, rendezvous_file);   // atomic publish
	} else {
		const auto t0 = std::chrono::steady_clock::now();
		for (;;) {
			std::ifstream f(rendezvous_file, std::ios::binary);
			std::string head(tag.size(), '\0');
			if (f.good() && f.read(head.data(), (std::streamsize)head.size()) && head == tag && f.read(reinterpret_cast<char*>(&id), sizeof(id))) break;
			if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) die("timed out waiting for rank 0's id", rendezvous_file.c_str());
			std::this_thread::sleep_for(std::chrono::milliseconds(50));
		}
	}
	ncclComm_t c;
	nccl_ok(ncclCommInitRank(&c, world_, id, rank_), "ncclCommInitRank");
	comm_ = c;
	hipStream_t s;
	hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
	stream_ = s;
	Barrier();
	if (rank_ == 0) std::filesystem::remove(rendezvous_file);   // every rank has joined: the id is spent
}

RankComm::~RankComm() {
	if (bounce_) (void)hipFree(bounce_);
	if (comm_) (void)ncclCommDestroy((ncclComm_t)comm_);
	if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
}

void RankComm::Barrier() {
	if (world_ <= 1) return;
	if (!bounce_) { bounce_ = DeviceAlloc(1024); bounce_count_ = 1024; }
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	nccl_ok(ncclAllReduce(bounce_, bounce_, 1, ncclFloat, ncclSum, (ncclComm_t)comm_, (hipStream_t)stream_), "ncclAllReduce");
	hip_ok(hipStreamSynchronize((hipStream_t)stream_), "barrier sync");
}

void RankComm::BroadcastDevice(float* dev, size_t count, int root) {
	if (world_ <= 1 || count == 0) return;
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	nccl_ok(ncclBroadcast(dev, dev, count, ncclFloat, root, (ncclComm_t)comm_, (hipStream_t)stream_), "ncclBroadcast");
	hip_ok(hipStreamSynchronize((hipStream_t)stream_), "broadcast sync");
}

void RankComm::BroadcastHost(void* host, size_t bytes, int root) {
	if (world_ <= 1 || bytes == 0) return;
	const size_t count = (bytes + 3) / 4;
	if (count > bounce_count_) {
		if (bounce_) (void)hipFree(bounce_);
		bounce_ = DeviceAlloc(count);
		bounce_count_ = count;
	}
	if (rank_ == root) hip_ok(hipMemcpy(bounce_, host, bytes, hipMemcpyHostToDevice), "stage in");
	BroadcastDevice(bounce_, count, root);
	if (rank_ != root) hip_ok(hipMemcpy(host, bounce_, bytes, hipMemcpyDeviceToHost), "stage out");
}

float* RankComm::DeviceAlloc(size_t count) {
	void* p = nullptr;
	hip_ok(hipMalloc(&p, (count ? count : 1) * sizeof(float)), "hipMalloc");
	return (float*)p;
}
void RankComm::DeviceFree(float* p) { if (p) (void)hipFree(p); }
void RankComm::HostToDevice(float* dev, const float* host, size_t count) { hip_ok(hipMemcpy(dev, host, count * 4, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
void RankComm::DeviceToHost(float* host, const float* dev, size_t count) { hip_ok(hipMemcpy(host, dev, count * 4, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
