// msc_comm.hpp -- the three collectives the sharded mean-shift driver needs (msc_sharded.hpp), behind one small interface:
//   RcclComm (msc_comm_rccl.hpp)  RCCL over xGMI, one rank per GPU: the production path; device buffers go straight to
//                                 ncclBroadcast / ncclAllGather / ncclAllReduce on the library's own HIP stream
//   TcpComm  (here)               plain sockets through rank 0, host memory: the bootstrap channel of RcclComm (it carries the
//                                 ncclUniqueId) and the transport of the tests -- several ranks sharing one GPU, or no GPU at all
//                                 (the CPU oracle as the rank-local scorer); device buffers are staged through host memory
// The reference has no counterpart (one process, OpenMP only: SURVEY section 5); the exchanges follow SURVEY 8(e).
// Ranks and rendezvous come from the environment torch.distributed.run sets: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR,
// MASTER_PORT (the sockets use MASTER_PORT + 1 + MSC_PORT_OFFSET: the launcher's own store listens on MASTER_PORT).
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace msc {

struct CommCalls { uint64_t broadcast = 0, all_gather = 0, all_reduce = 0, bytes = 0; };

struct Comm {
	int rank = 0, world = 1;
	CommCalls calls;
	// how a buffer flagged `dev` is reached from the host (set by whoever owns the device: msc_memcpy_to_host / _to_device / _device)
	std::function<void(void*, const void*, size_t)> to_host, to_device, on_device;
	virtual ~Comm() {}
	// every rank ends up with root's `bytes` at buf
	virtual void broadcast(void* buf, size_t bytes, int root, bool dev) = 0;
	// all = world ranges of bytes_each, rank-major; rank r contributes its first mine_bytes <= bytes_each (the rest of its range is
	// unspecified). mine may be all + rank * bytes_each (in place).
	virtual void all_gather(const void* mine, size_t mine_bytes, void* all, size_t bytes_each, bool dev) = 0;
	// element-wise sum of n uint64 over the ranks, in place
	virtual void all_reduce_sum_u64(void* buf, size_t n, bool dev) = 0;

	// host-side conveniences
	template <class T> std::vector<T> gather_values(const T& mine) {
		std::vector<T> all((size_t)world);
		all_gather(&mine, sizeof(T), all.data(), sizeof(T), false);
		return all;
	}
	uint64_t max_u64(uint64_t v) {
		uint64_t m = 0;
		for (uint64_t x : gather_values(v)) m = x > m ? x : m;
		return m;
	}
};

// one rank: nothing moves
struct SelfComm : Comm {
	void broadcast(void*, size_t, int, bool) override {}
	void all_gather(const void* mine, size_t mine_bytes, void* all, size_t, bool dev) override {
		if (mine == all || mine_bytes == 0) return;
		if (dev) on_device(all, mine, mine_bytes); else memcpy(all, mine, mine_bytes);
	}
	void all_reduce_sum_u64(void*, size_t, bool) override {}
};

struct CommEnv {
	int rank = 0, world = 1, local_rank = 0, port = 29517;
	std::string addr = "127.0.0.1";
	static CommEnv from_environment() {
		CommEnv e;
		auto geti = [](const char* n, int d) { const char* v = std::getenv(n); return v && *v ? std::atoi(v) : d; };
		e.rank = geti("RANK", 0);
		e.world = geti("WORLD_SIZE", 1);
		e.local_rank = geti("LOCAL_RANK", e.rank);
		if (const char* a = std::getenv("MASTER_ADDR")) if (*a) e.addr = a;
		e.port = geti("MASTER_PORT", 29500) + 1 + geti("MSC_PORT_OFFSET", 16);
		if (e.addr == "localhost") e.addr = "127.0.0.1";
		return e;
	}
};

class TcpComm : public Comm {
public:
	explicit TcpComm(const CommEnv& e, int timeout_s = 300) : timeout_s_(timeout_s) {
		rank = e.rank;
		world = e.world;
		if (world <= 1) return;
		if (rank == 0) {
			const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
			if (ls < 0) throw std::runtime_error("TcpComm: socket() failed");
			int one = 1;
			::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
			sockaddr_in a{};
			a.sin_family = AF_INET;
			a.sin_port = htons((uint16_t)e.port);
			a.sin_addr.s_addr = htonl(INADDR_ANY);
			if (::bind(ls, (sockaddr*)&a, sizeof a) != 0 || ::listen(ls, world) != 0) { ::close(ls); throw std::runtime_error("TcpComm: cannot listen on port " + std::to_string(e.port)); }
			timeval tv{timeout_s_, 0};
			::setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
			peers_.assign((size_t)world, -1);
			for (int i = 1; i < world; i++) {
				const int s = ::accept(ls, nullptr, nullptr);
				if (s < 0) { ::close(ls); throw std::runtime_error("TcpComm: a rank did not connect within the timeout"); }
				tune(s);
				int32_t r = -1;
				recv_all(s, &r, sizeof r);
				if (r <= 0 || r >= world || peers_[(size_t)r] != -1) { ::close(ls); throw std::runtime_error("TcpComm: unexpected rank at rendezvous"); }
				peers_[(size_t)r] = s;
			}
			::close(ls);
		} else {
			const auto t0 = std::chrono::steady_clock::now();
			for (;;) {
				const int s = ::socket(AF_INET, SOCK_STREAM, 0);
				sockaddr_in a{};
				a.sin_family = AF_INET;
				a.sin_port = htons((uint16_t)e.port);
				if (::inet_pton(AF_INET, e.addr.c_str(), &a.sin_addr) != 1) { ::close(s); throw std::runtime_error("TcpComm: MASTER_ADDR must be an IPv4 address"); }
				if (::connect(s, (sockaddr*)&a, sizeof a) == 0) { root_ = s; break; }
				::close(s);
				if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s_)) throw std::runtime_error("TcpComm: rank 0 is not listening");
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
			}
			tune(root_);
			const int32_t r = rank;
			send_all(root_, &r, sizeof r);
		}
	}
	~TcpComm() override {
		for (int s : peers_) if (s >= 0) ::close(s);
		if (root_ >= 0) ::close(root_);
	}

	void broadcast(void* buf, size_t bytes, int root, bool dev) override {
		if (world <= 1 || bytes == 0) return;
		calls.broadcast++;
		calls.bytes += bytes;
		uint8_t* h = host_view(buf, bytes, dev, rank == root);
		if (rank == 0) {
			if (root != 0) recv_all(peers_[(size_t)root], h, bytes);
			for (int r = 1; r < world; r++) if (r != root) send_all(peers_[(size_t)r], h, bytes);
		} else {
			if (rank == root) send_all(root_, h, bytes);
			else recv_all(root_, h, bytes);
		}
		if (dev && rank != root) to_device(buf, h, bytes);
	}

	void all_gather(const void* mine, size_t mine_bytes, void* all, size_t bytes_each, bool dev) override {
		if (mine_bytes > bytes_each) throw std::runtime_error("all_gather: contribution larger than its range");
		if (world <= 1) {
			if (mine != all && mine_bytes) { if (dev) on_device(all, mine, mine_bytes); else memcpy(all, mine, mine_bytes); }
			return;
		}
		calls.all_gather++;
		calls.bytes += bytes_each * (size_t)world;
		const size_t total = bytes_each * (size_t)world;
		stage_.resize(total);
		uint8_t* h = dev ? stage_.data() : (uint8_t*)all;
		uint8_t* my = h + (size_t)rank * bytes_each;
		if (mine_bytes) { if (dev) to_host(my, mine, mine_bytes); else if ((const void*)my != mine) memmove(my, mine, mine_bytes); }
		if (rank == 0) {
			for (int r = 1; r < world; r++) recv_all(peers_[(size_t)r], h + (size_t)r * bytes_each, bytes_each);
			for (int r = 1; r < world; r++) send_all(peers_[(size_t)r], h, total);
		} else {
			send_all(root_, my, bytes_each);
			recv_all(root_, h, total);
		}
		if (dev) to_device(all, h, total);
	}

	void all_reduce_sum_u64(void* buf, size_t n, bool dev) override {
		if (world <= 1 || n == 0) return;
		calls.all_reduce++;
		calls.bytes += n * 8;
		uint8_t* h = host_view(buf, n * 8, dev, true);
		uint64_t* v = (uint64_t*)h;
		if (rank == 0) {
			tmp_.resize(n);
			for (int r = 1; r < world; r++) {
				recv_all(peers_[(size_t)r], tmp_.data(), n * 8);
				for (size_t i = 0; i < n; i++) v[i] += tmp_[i];
			}
			for (int r = 1; r < world; r++) send_all(peers_[(size_t)r], v, n * 8);
		} else {
			send_all(root_, v, n * 8);
			recv_all(root_, v, n * 8);
		}
		if (dev) to_device(buf, h, n * 8);
	}

	// the receive / send deadline of every socket from now on (the wait for rank 0's trained model may take longer than a collective)
	void set_timeout(int seconds) {
		timeout_s_ = seconds;
		for (int s : peers_) if (s >= 0) tune(s);
		if (root_ >= 0) tune(root_);
	}

private:
	uint8_t* host_view(void* buf, size_t bytes, bool dev, bool fill) {
		if (!dev) return (uint8_t*)buf;
		stage_.resize(bytes);
		if (fill) to_host(stage_.data(), buf, bytes);
		return stage_.data();
	}
	void tune(int s) const {
		int one = 1;
		::setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
		timeval tv{timeout_s_, 0};
		::setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
		::setsockopt(s, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
	}
	static void send_all(int s, const void* p, size_t n) {
		const uint8_t* b = (const uint8_t*)p;
		while (n) {
			const ssize_t w = ::send(s, b, n, MSG_NOSIGNAL);
			if (w <= 0) { if (w < 0 && errno == EINTR) continue; throw std::runtime_error("TcpComm: a peer went away (send)"); }
			b += w; n -= (size_t)w;
		}
	}
	static void recv_all(int s, void* p, size_t n) {
		uint8_t* b = (uint8_t*)p;
		while (n) {
			const ssize_t r = ::recv(s, b, n, 0);
			if (r <= 0) { if (r < 0 && errno == EINTR) continue; throw std::runtime_error("TcpComm: a peer went away or timed out (recv)"); }
			b += r; n -= (size_t)r;
		}
	}
	int timeout_s_;
	std::vector<int> peers_;          // rank 0: socket of every other rank
	int root_ = -1;                   // others: socket to rank 0
	std::vector<uint8_t> stage_;
	std::vector<uint64_t> tmp_;
};

}  // namespace msc
