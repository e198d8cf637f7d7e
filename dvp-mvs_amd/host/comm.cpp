// comm.cpp — RankComm on RCCL (ncclBroadcast / ncclAllReduce over xGMI) or on host-staged TCP, see comm.h.
#define __HIP_PLATFORM_AMD__ 1
#include <string>
#include <cctype>
#include <sched.h>
#include "comm.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <thread>

namespace {
RankComm* g_current = nullptr;
using Clock = std::chrono::steady_clock;

[[noreturn]] void die_local(const char* what, const char* detail) {
	fprintf(stderr, "RankComm: %s: %s\n", what, detail);
	if (g_current) g_current->Abort(what);
	exit(EXIT_FAILURE);
}
void hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) die_local(what, hipGetErrorString(e)); }
void nccl_ok(ncclResult_t r, const char* what) { if (r != ncclSuccess) die_local(what, ncclGetErrorString(r)); }

// the rendezvous record: tag line (with the run's nonce) + payload
bool read_record(const std::string& file, const std::string& tag, void* payload, size_t bytes, double not_older_than_s) {
	std::error_code ec;
	const auto mt = std::filesystem::last_write_time(file, ec);
	if (ec) return false;
	const double age = std::chrono::duration<double>(std::filesystem::file_time_type::clock::now() - mt).count();
	if (age > not_older_than_s) return false;   // left by an earlier run
	std::ifstream f(file, std::ios::binary);
	std::string head(tag.size(), '\0');
	return f.good() && f.read(head.data(), (std::streamsize)head.size()) && head == tag && f.read(reinterpret_cast<char*>(payload), (std::streamsize)bytes);
}
void publish_record(const std::string& file, const std::string& tag, const void* payload, size_t bytes) {
	const std::string tmp = file + ".tmp";
	{
		std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
		f.write(tag.data(), (std::streamsize)tag.size());
		f.write(reinterpret_cast<const char*>(payload), (std::streamsize)bytes);
		if (!f.good()) die_local("cannot write", tmp.c_str());
	}
	std::filesystem::rename(tmp, file);   // atomic publish
}
}  // namespace

RankComm* RankComm::Current() { return g_current; }

RankComm::RankComm(int rank, int world, int device, const std::string& rendezvous_file, const std::string& nonce, int timeout_s, const std::string& transport)
	: rank_(rank), world_(world), device_(device), timeout_s_(timeout_s), host_transport_(transport == "host"), abort_marker_(rendezvous_file + ".abort") {
	const auto started = Clock::now();
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	if (world_ <= 1) return;
	if (transport != "rccl" && transport != "host") die_local("unknown transport (rccl | host)", transport.c_str());
	if (nonce.empty()) die_local("--job <id unique to this run> is required with --world > 1", "a stale rendezvous file of an earlier run must never match");
	if (rank_ < 0 || rank_ >= world_) die_local("rank out of range", std::to_string(rank_).c_str());
	const std::string tag = "dvp-" + transport + "-id " + nonce + "\n";
	std::error_code ec;
	if (rank_ == 0) {   // nothing of an earlier run may survive into this one
		std::filesystem::remove(rendezvous_file, ec);
		std::filesystem::remove(rendezvous_file + ".tmp", ec);
		std::filesystem::remove(abort_marker_, ec);
	}
	g_current = this;
	auto waited = [&]() { return std::chrono::duration<double>(Clock::now() - started).count(); };
	if (host_transport_) {
		peers_.assign(rank_ == 0 ? (size_t)world_ : 1u, -1);
		if (rank_ == 0) {
			listen_fd_ = socket(AF_INET, SOCK_STREAM, 0);
			sockaddr_in a{};
			a.sin_family = AF_INET;
			a.sin_addr.s_addr = htonl(INADDR_ANY);
			a.sin_port = 0;
			int one = 1;
			setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
			if (listen_fd_ < 0 || bind(listen_fd_, (sockaddr*)&a, sizeof(a)) != 0 || listen(listen_fd_, world_) != 0) die_local("host transport", "cannot listen");
			socklen_t len = sizeof(a);
			getsockname(listen_fd_, (sockaddr*)&a, &len);
			const char* adv = std::getenv("DVP_HOST_ADDR");   // address the other ranks reach rank 0 at
			char rec[64] = { 0 };
			snprintf(rec, sizeof(rec), "%s:%d", adv ? adv : "127.0.0.1", (int)ntohs(a.sin_port));
			publish_record(rendezvous_file, tag, rec, sizeof(rec));
			for (int joined = 1; joined < world_;) {
				pollfd p{ listen_fd_, POLLIN, 0 };
				if (poll(&p, 1, 200) > 0) {
					const int fd = accept(listen_fd_, nullptr, nullptr);
					int32_t r = -1;
					if (fd >= 0 && recv(fd, &r, 4, MSG_WAITALL) == 4 && r > 0 && r < world_ && peers_[(size_t)r] < 0) { peers_[(size_t)r] = fd; ++joined; }
					else if (fd >= 0) close(fd);
				}
				if (waited() > timeout_s_) die_local("timed out waiting for the other ranks to connect", rendezvous_file.c_str());
			}
		} else {
			char rec[64] = { 0 };
			while (!read_record(rendezvous_file, tag, rec, sizeof(rec), waited() + 120.0)) {
				if (waited() > timeout_s_) die_local("timed out waiting for rank 0's address", rendezvous_file.c_str());
				std::this_thread::sleep_for(std::chrono::milliseconds(50));
			}
			std::string s(rec);
			const size_t colon = s.rfind(':');
			sockaddr_in a{};
			a.sin_family = AF_INET;
			a.sin_port = htons((uint16_t)std::atoi(s.substr(colon + 1).c_str()));
			inet_pton(AF_INET, s.substr(0, colon).c_str(), &a.sin_addr);
			const int fd = socket(AF_INET, SOCK_STREAM, 0);
			if (fd < 0 || connect(fd, (sockaddr*)&a, sizeof(a)) != 0) die_local("cannot connect to rank 0", rec);
			const int32_t r = rank_;
			SendAll(fd, &r, 4);
			peers_[0] = fd;
		}
		for (int fd : peers_)
			if (fd >= 0) { int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }
	} else {
		ncclUniqueId id;
		if (rank_ == 0) {
			nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
			publish_record(rendezvous_file, tag, &id, sizeof(id));
		} else {
			while (!read_record(rendezvous_file, tag, &id, sizeof(id), waited() + 120.0)) {
				if (waited() > timeout_s_) die_local("timed out waiting for rank 0's id", rendezvous_file.c_str());
				std::this_thread::sleep_for(std::chrono::milliseconds(50));
			}
		}
		ncclComm_t c;
		nccl_ok(ncclCommInitRank(&c, world_, id, rank_), "ncclCommInitRank");
		comm_ = c;
		hipStream_t s;
		hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
		stream_ = s;
	}
	Barrier();
	if (rank_ == 0) std::filesystem::remove(rendezvous_file, ec);   // every rank has joined: the record is spent
}

RankComm::~RankComm() {
	if (g_current == this) g_current = nullptr;
	if (bounce_) (void)hipFree(bounce_);
	if (comm_) (void)ncclCommDestroy((ncclComm_t)comm_);
	if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
	for (int fd : peers_) if (fd >= 0) close(fd);
	if (listen_fd_ >= 0) close(listen_fd_);
}

bool RankComm::PeerAborted() const {
	std::error_code ec;
	return world_ > 1 && std::filesystem::exists(abort_marker_, ec);
}

void RankComm::Abort(const char* why) {
	static bool once = false;
	if (!once && world_ > 1) {
		once = true;
		fprintf(stderr, "RankComm[rank %d]: aborting the job: %s\n", rank_, why);
		std::ofstream(abort_marker_, std::ios::app) << "rank " << rank_ << ": " << why << "\n";
		if (comm_) (void)ncclCommAbort((ncclComm_t)comm_);   // releases this rank from any collective in flight
		comm_ = nullptr;
		for (int fd : peers_) if (fd >= 0) close(fd);
		peers_.clear();
	}
	fflush(nullptr);
	_exit(EXIT_FAILURE);
}

// Wait for the collective(s) queued on the stream WITHOUT blocking in the runtime: a peer that died or
// aborted is noticed through the marker file / RCCL's asynchronous error / the timeout.
void RankComm::WaitStream(const char* what) {
	const auto t0 = Clock::now();
	int spin = 0;
	for (;;) {
		const hipError_t q = hipStreamQuery((hipStream_t)stream_);
		if (q == hipSuccess) return;
		if (q != hipErrorNotReady) die_local(what, hipGetErrorString(q));
		if (++spin < 2000) continue;                    // ~first 100 us: pure polling (small collectives finish here)
		std::this_thread::sleep_for(std::chrono::microseconds(200));
		if (spin % 512 == 0) {
			ncclResult_t async = ncclSuccess;
			if (comm_ && ncclCommGetAsyncError((ncclComm_t)comm_, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) Abort(ncclGetErrorString(async));
			if (PeerAborted()) Abort("another rank aborted (see the .abort marker)");
			if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s_) Abort("collective timed out (a peer died?)");
		}
	}
}

void RankComm::SendAll(int fd, const void* p, size_t n) {
	const char* c = static_cast<const char*>(p);
	while (n) {
		const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
		if (k <= 0) Abort("host transport: peer closed the connection (send)");
		c += k;
		n -= (size_t)k;
	}
}
void RankComm::RecvAll(int fd, void* p, size_t n) {
	char* c = static_cast<char*>(p);
	const auto t0 = Clock::now();
	while (n) {
		pollfd pf{ fd, POLLIN, 0 };
		const int pr = poll(&pf, 1, 200);
		if (pr == 0) {
			if (PeerAborted()) Abort("another rank aborted (see the .abort marker)");
			if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s_) Abort("host transport: receive timed out (a peer died?)");
			continue;
		}
		const ssize_t k = recv(fd, c, n, 0);
		if (k <= 0) Abort("host transport: peer closed the connection (recv)");
		c += k;
		n -= (size_t)k;
	}
}
// star through rank 0: the root hands the buffer to rank 0 (unless it is rank 0), rank 0 sends it to everyone else
void RankComm::HostBroadcast(void* buf, size_t bytes, int root) {
	if (rank_ == 0) {
		if (root != 0) RecvAll(peers_[(size_t)root], buf, bytes);
		for (int r = 1; r < world_; ++r)
			if (r != root) SendAll(peers_[(size_t)r], buf, bytes);
	} else if (rank_ == root) {
		SendAll(peers_[0], buf, bytes);
	} else {
		RecvAll(peers_[0], buf, bytes);
	}
}

bool RankComm::AllOk(bool ok) {
	if (world_ <= 1) return ok;
	float v = ok ? 1.0f : 0.0f;
	if (host_transport_) {
		if (rank_ == 0) {
			for (int r = 1; r < world_; ++r) { float o; RecvAll(peers_[(size_t)r], &o, 4); v = (o < v) ? o : v; }
			for (int r = 1; r < world_; ++r) SendAll(peers_[(size_t)r], &v, 4);
		} else {
			SendAll(peers_[0], &v, 4);
			RecvAll(peers_[0], &v, 4);
		}
		return v > 0.5f;
	}
	if (!bounce_) { bounce_ = DeviceAlloc(1024); bounce_count_ = 1024; }
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	hip_ok(hipMemcpy(bounce_, &v, 4, hipMemcpyHostToDevice), "stage in");
	nccl_ok(ncclAllReduce(bounce_, bounce_, 1, ncclFloat, ncclMin, (ncclComm_t)comm_, (hipStream_t)stream_), "ncclAllReduce");
	WaitStream("all-reduce");
	hip_ok(hipMemcpy(&v, bounce_, 4, hipMemcpyDeviceToHost), "stage out");
	return v > 0.5f;
}

void RankComm::Barrier() { (void)AllOk(true); }

void RankComm::BroadcastDevice(float* dev, size_t count, int root) {
	if (world_ <= 1 || count == 0) return;
	hip_ok(hipSetDevice(device_), "hipSetDevice");
	if (host_transport_) {
		stage_.resize(count * 4);
		if (rank_ == root) hip_ok(hipMemcpy(stage_.data(), dev, count * 4, hipMemcpyDeviceToHost), "stage out");
		HostBroadcast(stage_.data(), count * 4, root);
		if (rank_ != root) hip_ok(hipMemcpy(dev, stage_.data(), count * 4, hipMemcpyHostToDevice), "stage in");
		return;
	}
	nccl_ok(ncclBroadcast(dev, dev, count, ncclFloat, root, (ncclComm_t)comm_, (hipStream_t)stream_), "ncclBroadcast");
	WaitStream("broadcast");
}

void RankComm::BroadcastHost(void* host, size_t bytes, int root) {
	if (world_ <= 1 || bytes == 0) return;
	if (host_transport_) { HostBroadcast(host, bytes, root); return; }
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

int RankComm::BindProcessNearDevice(int device) {
	if (std::getenv("DVP_NO_NUMA_BIND")) return -1;
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
	std::string id(bus);
	for (char& ch : id) ch = (char)std::tolower((unsigned char)ch);
	int node = -1;
	{
		std::ifstream f("/sys/bus/pci/devices/" + id + "/numa_node");
		if (!(f >> node) || node < 0) return -1;
	}
	std::string list;
	{
		std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
		if (!std::getline(f, list) || list.empty()) return -1;
	}
	cpu_set_t want, have, both;
	CPU_ZERO(&want);
	for (size_t i = 0; i < list.size();) {   // "0-63,128-191"
		size_t j = i;
		long a = std::strtol(list.c_str() + i, nullptr, 10), b = a;
		while (j < list.size() && list[j] != ',' && list[j] != '-') ++j;
		if (j < list.size() && list[j] == '-') { b = std::strtol(list.c_str() + j + 1, nullptr, 10); while (j < list.size() && list[j] != ',') ++j; }
		for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &want);
		i = j + 1;
	}
	if (sched_getaffinity(0, sizeof(have), &have) != 0) return -1;
	CPU_AND(&both, &want, &have);
	if (CPU_COUNT(&both) == 0 || sched_setaffinity(0, sizeof(both), &both) != 0) return -1;
	return node;
}
void RankComm::BindThisThread(int device) { hip_ok(hipSetDevice(device), "hipSetDevice"); }
float* RankComm::DeviceAlloc(size_t count) {
	void* p = nullptr;
	hip_ok(hipMalloc(&p, (count ? count : 1) * sizeof(float)), "hipMalloc");
	return (float*)p;
}
void RankComm::DeviceFree(float* p) { if (p) (void)hipFree(p); }
void RankComm::HostToDevice(float* dev, const float* host, size_t count) { hip_ok(hipMemcpy(dev, host, count * 4, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
void RankComm::DeviceToHost(float* host, const float* dev, size_t count) { hip_ok(hipMemcpy(host, dev, count * 4, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
