// comm.h — the multi-GPU side of the driver: one process per GPU, RCCL over xGMI.
//
// Views of a pass are independent, so ranks share nothing on the data path of a view.
// What IS shared: the input images of a pyramid level (every rank decodes its share, ncclBroadcast to every
// GPU) and, between passes, the new depth maps (the geometric-consistency term of the next pass reads every
// source view's depth map of the previous pass, APD.cpp:1147-1166): the owner of a view broadcasts its map,
// every rank keeps all maps resident on its device.  The reference has one device per process and no
// exchange at all (main.cpp:430-434); its inter-pass API is the result files.
//
// Failure model.  A rank that fails (unreadable image, size mismatch, write error, HIP/RCCL error) calls
// Abort(): it drops the marker file `<rendezvous>.abort` into the shared folder, ncclCommAbort()s its own
// communicator and exits non-zero.  Every wait of every rank polls the stream instead of blocking: it sees
// the marker (or an asynchronous RCCL error, or the per-collective timeout) within ~0.2 s and aborts too, so
// no rank is left hanging inside a collective.  AllOk() is the agreed check in front of the per-pass exchange.
//
// Transports.  "rccl" (default): ncclBroadcast / ncclAllReduce on device buffers.  "host": the same calls
// staged through host memory over TCP (star through rank 0) — for nodes without a working xGMI fabric and
// for the 2-rank driver test on a 1-GPU box (RCCL refuses two ranks on one device); the sharding, exchange,
// rendezvous and abort logic above it is the same code.
#ifndef DVP_HOST_COMM_H_
#define DVP_HOST_COMM_H_
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

class RankComm {
public:
	// world == 1: no communicator is created, every call is a local no-op / copy.
	// The ncclUniqueId (transport "host": rank 0's TCP port) travels from rank 0 to the others through
	// `rendezvous_file` (shared folder), written atomically.  `nonce` must be the same on all ranks and
	// unique per run (the launcher's job id): rank 0 deletes any file left by an earlier run before it
	// publishes, the others ignore files with another nonce or older than their own start by > 120 s.
	RankComm(int rank, int world, int device, const std::string& rendezvous_file, const std::string& nonce,
	         int timeout_s = 300, const std::string& transport = "rccl");
	~RankComm();
	int rank() const { return rank_; }
	int world() const { return world_; }
	void Barrier();
	bool AllOk(bool ok);   // logical AND over ranks (collective)
	// device buffers (on this rank's GPU); count in floats
	void BroadcastDevice(float* dev, size_t count, int root);
	// host buffers (transport rccl: staged through a device bounce buffer)
	void BroadcastHost(void* host, size_t bytes, int root);
	// agreed shutdown: marker file + ncclCommAbort + exit(EXIT_FAILURE).  Safe to call from any failure path.
	[[noreturn]] void Abort(const char* why);
	// the communicator Abort()/the fatal hook act on (the driver has one)
	static RankComm* Current();
	// device memory helpers so that callers need no HIP headers
	// Confines the calling process to the CPUs of the NUMA node the device hangs off (sysfs: the PCI device's numa_node, the
	// node's cpulist), so that its threads, its first-touched memory and the runtime's pinned buffers sit next to the GPU: on a
	// two-socket host a process that started on the far socket moved a view's maps at a third of the rate.  No-op when the
	// topology is not visible or DVP_NO_NUMA_BIND is set.  Returns the node (or -1).
	static int BindProcessNearDevice(int device);
	static void BindThisThread(int device);   // hipSetDevice for a driver thread other than main's (HIP's current device is per thread)
	static float* DeviceAlloc(size_t count);
	static void DeviceFree(float* p);
	static void HostToDevice(float* dev, const float* host, size_t count);
	static void DeviceToHost(float* host, const float* dev, size_t count);
private:
	void WaitStream(const char* what);
	bool PeerAborted() const;
	void HostBroadcast(void* buf, size_t bytes, int root);
	void SendAll(int fd, const void* p, size_t n);
	void RecvAll(int fd, void* p, size_t n);
	int rank_, world_, device_;
	int timeout_s_;
	bool host_transport_ = false;
	std::string abort_marker_;
	void* comm_ = nullptr;     // ncclComm_t
	void* stream_ = nullptr;   // hipStream_t
	float* bounce_ = nullptr;
	size_t bounce_count_ = 0;
	std::vector<int> peers_;   // host transport: rank 0 holds one socket per rank (index = rank), the others hold [0]
	int listen_fd_ = -1;
	std::vector<char> stage_;
};
#endif
