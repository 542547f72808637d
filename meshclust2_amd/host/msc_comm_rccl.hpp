// msc_comm_rccl.hpp -- msc::Comm over RCCL (rccl.h): one rank per GPU, collectives over xGMI, queued on the HIP stream of the
// rank's msc_ctx (msc_stream_handle) so that they are ordered with the library's kernels without host round trips. Device buffers
// (the packed query, the gathered column sums, the packed centres) go to ncclBroadcast / ncclAllGather / ncclAllReduce as they
// are; the few host-side records (24-byte get_close records, (distance, position) pairs) are staged through a small device buffer.
// The ncclUniqueId travels over the TcpComm given at construction. SURVEY 8(e); the reference has no counterpart.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "../../include/meshclust2_hip.h"
#include "msc_comm.hpp"

namespace msc {

// Failure handling: the communicator is initialised NON-BLOCKING (ncclCommInitRankConfig, blocking = 0) and every call is
// followed by a poll of ncclCommGetAsyncError against a deadline; a watchdog thread watches the stream behind the last collective. A
// rank that crashed or left through _Exit therefore cannot leave its peers inside ncclBroadcast / ncclAllGather (or in the library's
// hipStreamSynchronize behind them) for ever: on an error or a timeout (MSC_COMM_TIMEOUT_S, default 300 s, the TcpComm's figure) the
// communicator is aborted (ncclCommAbort) and the process ends non-zero -- no retry inside the same process.
// MSC_FORCE_SHARDED (msc_cluster with WORLD_SIZE = 1) disables the one-rank shortcuts below, so that communicator set-up, the in-place
// all-gather, the broadcast, the uint64 all-reduce and the host staging all execute on a single GPU (tests/test_gpu_parity.py).
class RcclComm : public Comm {
public:
	// `ctx` must be the context whose sets hold the buffers; boot is only used here
	RcclComm(msc_ctx* ctx, Comm& boot) : ctx_(ctx) {
		rank = boot.rank;
		world = boot.world;
		force_ = std::getenv("MSC_FORCE_SHARDED") != nullptr;
		if (const char* e = std::getenv("MSC_COMM_TIMEOUT_S")) if (std::atoi(e) > 0) timeout_s_ = std::atoi(e);
		stream_ = (hipStream_t)msc_stream_handle(ctx);
		if (hipGetDevice(&device_) != hipSuccess) device_ = 0;
		ncclUniqueId id;
		memset(&id, 0, sizeof id);
		if (rank == 0) check(ncclGetUniqueId(&id), "ncclGetUniqueId");
		boot.broadcast(&id, sizeof id, 0, false);
		ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
		cfg.blocking = 0;
		const ncclResult_t r = ncclCommInitRankConfig(&comm_, world, id, rank, &cfg);
		if (r != ncclSuccess && r != ncclInProgress) check(r, "ncclCommInitRankConfig");
		settle("ncclCommInitRankConfig");
		if (hipEventCreateWithFlags(&ev_, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("RcclComm: hipEventCreate failed");
		watchdog_ = std::thread([this] { watch(); });
	}
	~RcclComm() override {
		stop_ = true;
		if (watchdog_.joinable()) watchdog_.join();
		if (comm_) {          // a non-blocking communicator: finalize, wait for it (bounded), then destroy
			(void)ncclCommFinalize(comm_);
			ncclResult_t st = ncclInProgress;
			for (int i = 0; i < 3000 && ncclCommGetAsyncError(comm_, &st) == ncclSuccess && st == ncclInProgress; i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
			(void)ncclCommDestroy(comm_);
		}
		if (ev_) (void)hipEventDestroy(ev_);
		if (scratch_) (void)msc_device_free(ctx_, scratch_);
	}

	void broadcast(void* buf, size_t bytes, int root, bool dev) override {
		if ((world <= 1 && !force_) || bytes == 0) return;
		calls.broadcast++;
		calls.bytes += bytes;
		void* d = dev ? buf : staged(bytes, rank == root ? buf : nullptr, bytes);
		issue(ncclBroadcast(d, d, bytes, ncclUint8, root, comm_, stream_), "ncclBroadcast");
		if (!dev) unstage(buf, d, bytes);
	}
	void all_gather(const void* mine, size_t mine_bytes, void* all, size_t bytes_each, bool dev) override {
		if (mine_bytes > bytes_each) throw std::runtime_error("all_gather: contribution larger than its range");
		const size_t total = bytes_each * (size_t)world;
		if (world <= 1 && !force_) {
			if (mine != all && mine_bytes) { if (dev) on_device(all, mine, mine_bytes); else memcpy(all, mine, mine_bytes); }
			return;
		}
		calls.all_gather++;
		calls.bytes += total;
		uint8_t* d = dev ? (uint8_t*)all : (uint8_t*)staged(total, nullptr, 0);
		uint8_t* my = d + (size_t)rank * bytes_each;
		if (mine_bytes) {
			if (dev) { if ((const void*)my != mine) on_device(my, mine, mine_bytes); }
			else to_device(my, mine, mine_bytes);
		}
		issue(ncclAllGather(my, d, bytes_each, ncclUint8, comm_, stream_), "ncclAllGather");      // in place: my == d + rank * count
		if (!dev) unstage(all, d, total);
	}
	void all_reduce_sum_u64(void* buf, size_t n, bool dev) override {
		if ((world <= 1 && !force_) || n == 0) return;
		calls.all_reduce++;
		calls.bytes += n * 8;
		void* d = dev ? buf : staged(n * 8, buf, n * 8);
		issue(ncclAllReduce(d, d, n, ncclUint64, ncclSum, comm_, stream_), "ncclAllReduce");
		if (!dev) unstage(buf, d, n * 8);
	}

private:
	[[noreturn]] void give_up(const char* what, const char* why) {
		std::fprintf(stderr, "rank %d: %s %s: aborting the communicator\n", rank, what, why);
		if (comm_) (void)ncclCommAbort(comm_);
		std::_Exit(4);          // (no destructors: the other ranks must see this one go)
	}
	void check(ncclResult_t r, const char* what) const {
		if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
	}
	// a non-blocking communicator answers ncclInProgress until the call has been taken: poll it against the deadline
	void settle(const char* what) {
		const auto t0 = std::chrono::steady_clock::now();
		for (uint64_t spin = 0;; spin++) {
			ncclResult_t st = ncclSuccess;
			const ncclResult_t q = ncclCommGetAsyncError(comm_, &st);
			if (q != ncclSuccess) give_up(what, ncclGetErrorString(q));
			if (st == ncclSuccess) return;
			if (st != ncclInProgress) give_up(what, ncclGetErrorString(st));
			if (spin < 2000) continue;          // (an enqueue is taken within microseconds: no sleep on the step's critical path)
			if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s_)) give_up(what, "timed out");
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		}
	}
	void issue(ncclResult_t r, const char* what) {
		if (r != ncclSuccess && r != ncclInProgress) give_up(what, ncclGetErrorString(r));
		settle(what);
		// the collective is queued on the stream: the watchdog follows the event behind it
		(void)hipEventRecord(ev_, stream_);
		pending_since_.store(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
	}
	void watch() {
		(void)hipSetDevice(device_);
		while (!stop_) {
			std::this_thread::sleep_for(std::chrono::milliseconds(100));
			const int64_t since = pending_since_.load();
			if (!since) continue;
			if (hipEventQuery(ev_) == hipSuccess) {          // (a newer collective re-arms the stamp when it is issued)
				int64_t expect = since;
				pending_since_.compare_exchange_strong(expect, 0);
				continue;
			}
			const int64_t now = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
			if (now - since > (int64_t)timeout_s_ * 1000) give_up("a collective", "did not complete within the deadline (a peer is gone?)");
		}
	}
	void* staged(size_t bytes, const void* fill, size_t fill_bytes) {
		if (bytes > scratch_cap_) {
			if (scratch_) (void)msc_device_free(ctx_, scratch_);
			scratch_ = nullptr;
			scratch_cap_ = bytes + bytes / 2 + 4096;
			if (msc_device_malloc(ctx_, scratch_cap_, &scratch_) != MSC_OK) throw std::runtime_error("RcclComm: out of device memory for staging");
		}
		if (fill && fill_bytes) to_device(scratch_, fill, fill_bytes);
		return scratch_;
	}
	void unstage(void* host, const void* d, size_t bytes) { to_host(host, d, bytes); }      // (msc_memcpy_to_host drains the stream)

	msc_ctx* ctx_;
	hipStream_t stream_ = nullptr;
	ncclComm_t comm_ = nullptr;
	void* scratch_ = nullptr;
	size_t scratch_cap_ = 0;
	bool force_ = false;
	int timeout_s_ = 300;
	int device_ = 0;
	hipEvent_t ev_ = nullptr;
	std::atomic<int64_t> pending_since_{0};
	std::atomic<bool> stop_{false};
	std::thread watchdog_;
};

}  // namespace msc
