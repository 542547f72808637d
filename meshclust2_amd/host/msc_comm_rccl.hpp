// msc_comm_rccl.hpp -- msc::Comm over RCCL (rccl.h): one rank per GPU, collectives over xGMI, queued on the HIP stream of the
// rank's msc_ctx (msc_stream_handle) so that they are ordered with the library's kernels without host round trips. Device buffers
// (the packed query, the gathered column sums, the packed centres) go to ncclBroadcast / ncclAllGather / ncclAllReduce as they
// are; the few host-side records (24-byte get_close records, (distance, position) pairs) are staged through a small device buffer.
// The ncclUniqueId travels over the TcpComm given at construction. SURVEY 8(e); the reference has no counterpart.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/meshclust2_hip.h"
#include "msc_comm.hpp"

namespace msc {

class RcclComm : public Comm {
public:
	// `ctx` must be the context whose sets hold the buffers; boot is only used here
	RcclComm(msc_ctx* ctx, Comm& boot) : ctx_(ctx) {
		rank = boot.rank;
		world = boot.world;
		stream_ = (hipStream_t)msc_stream_handle(ctx);
		ncclUniqueId id;
		memset(&id, 0, sizeof id);
		if (rank == 0) check(ncclGetUniqueId(&id), "ncclGetUniqueId");
		boot.broadcast(&id, sizeof id, 0, false);
		check(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
	}
	~RcclComm() override {
		if (comm_) (void)ncclCommDestroy(comm_);
		if (scratch_) (void)msc_device_free(ctx_, scratch_);
	}

	void broadcast(void* buf, size_t bytes, int root, bool dev) override {
		if (world <= 1 || bytes == 0) return;
		calls.broadcast++;
		calls.bytes += bytes;
		void* d = dev ? buf : staged(bytes, rank == root ? buf : nullptr, bytes);
		check(ncclBroadcast(d, d, bytes, ncclUint8, root, comm_, stream_), "ncclBroadcast");
		if (!dev) unstage(buf, d, bytes);
	}
	void all_gather(const void* mine, size_t mine_bytes, void* all, size_t bytes_each, bool dev) override {
		if (mine_bytes > bytes_each) throw std::runtime_error("all_gather: contribution larger than its range");
		const size_t total = bytes_each * (size_t)world;
		if (world <= 1) {
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
		check(ncclAllGather(my, d, bytes_each, ncclUint8, comm_, stream_), "ncclAllGather");      // in place: my == d + rank * count
		if (!dev) unstage(all, d, total);
	}
	void all_reduce_sum_u64(void* buf, size_t n, bool dev) override {
		if (world <= 1 || n == 0) return;
		calls.all_reduce++;
		calls.bytes += n * 8;
		void* d = dev ? buf : staged(n * 8, buf, n * 8);
		check(ncclAllReduce(d, d, n, ncclUint64, ncclSum, comm_, stream_), "ncclAllReduce");
		if (!dev) unstage(buf, d, n * 8);
	}

private:
	void check(ncclResult_t r, const char* what) const {
		if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
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
};

}  // namespace msc
