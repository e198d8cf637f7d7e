// comm.h — the multi-GPU side of the driver: one process per GPU, RCCL over xGMI.
//
// Views of a pass are independent, so ranks share nothing on the data path of a view.
// What IS shared: the input images of a pyramid level (decoded and resized once, by rank 0, then
// ncclBroadcast to every GPU) and, between passes, the new depth maps (the geometric-consistency term of
// the next pass reads every source view's depth map of the previous pass, APD.cpp:1147-1166): the owner
// of a view broadcasts its map, every rank keeps all maps resident on its device.  The reference has one
// device per process and no exchange at all (main.cpp:430-434); its inter-pass API is the result files.
#ifndef DVP_HOST_COMM_H_
#define DVP_HOST_COMM_H_
#include <cstddef>
#include <cstdint>
#include <string>

class RankComm {
public:
	// world == 1: no RCCL communicator is created, every call is a local no-op / copy.
	// The ncclUniqueId travels from rank 0 to the others through `rendezvous_file` (shared folder), written
	// atomically; `nonce` (same on all ranks, e.g. the launcher's job id) keeps a stale file of an earlier
	// run from being picked up.  Exits with a message on failure or after `timeout_s`.
	RankComm(int rank, int world, int device, const std::string& rendezvous_file, const std::string& nonce, int timeout_s = 300);
	~RankComm();
	int rank() const { return rank_; }
	int world() const { return world_; }
	void Barrier();
	// device buffers (on this rank's GPU); count in floats
	void BroadcastDevice(float* dev, size_t count, int root);
	// host buffers staged through a device bounce buffer
	void BroadcastHost(void* host, size_t bytes, int root);
	// device memory helpers so that callers need no HIP headers
	static float* DeviceAlloc(size_t count);
	static void DeviceFree(float* p);
	static void HostToDevice(float* dev, const float* host, size_t count);
	static void DeviceToHost(float* host, const float* dev, size_t count);
private:
	int rank_, world_, device_;
	void* comm_ = nullptr;     // ncclComm_t
	void* stream_ = nullptr;   // hipStream_t
	float* bounce_ = nullptr;
	size_t bounce_count_ = 0;
};
#endif
