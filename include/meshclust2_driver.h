/* include/meshclust2_driver.h
 *
 * C entry points of libmsc_driver.so: MeShClust2's mean-shift clustering LOGIC (ClusterFactory::MS /
 * accumulate / mean_shift_update / merge / print_output, cluster/ClusterFactory.cpp:288-435,553-656; bvec,
 * cluster/bvec.cpp; the sorts and ids of do_run, cluster/CRunner.cpp:538-539,574-597) with the hot path behind callbacks.
 * Pure host code (no HIP): the same logic drives one GPU (meshclust2_amd/host/msc_cluster links it as C++), one GPU per rank
 * (meshclust2_amd/cluster.py: every rank runs it on replicated flags and lists, its callbacks shard the scoring over
 * libmeshclust2_hip.so and exchange results over torch.distributed) and the CPU oracle in the world-size-2 gloo test.
 *
 * Points are handles 0..n-1 (position in the arrays given to msc_cluster_run), centres are handles the caller hands out in
 * centre_new. Every callback returns 0 or a non-zero status that aborts the run (msc_cluster_run then returns it).
 */
#ifndef MESHCLUST2_DRIVER_H
#define MESHCLUST2_DRIVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	void* user;
	/* Trainer::get_close(last, window, is_min) (cluster/Trainer.cpp:23-71; caller cluster/ClusterFactory.cpp:566): query = point q,
	 * window[0..m) in window order; flags[j] = 1 where the candidate is marked close, *pos = position in the window of the first
	 * arg-max of combo 0 or -1, *is_min = no candidate was close */
	int (*get_close)(void* user, uint32_t q, const uint32_t* window, uint64_t m, uint8_t* flags, int64_t* pos, int* is_min);
	/* get_mean / Trainer::closest (cluster/ClusterFactory.cpp:338-380; cluster/Trainer.cpp:144-157): *pos = first member nearest the mean */
	int (*closest)(void* user, const uint32_t* members, uint64_t m, int64_t* pos);
	/* Center(c): center(c->clone()) (cluster/Center.h:13-40; cluster/ClusterFactory.cpp:603) */
	int (*centre_new)(void* user, uint32_t point, uint32_t* centre);
	/* center->set(*next) (cluster/ClusterFactory.cpp:328,331): bins, length, id -- not mag */
	int (*centre_set)(void* user, uint32_t centre, uint32_t point);
	/* Trainer::filter(centre, points) (cluster/Trainer.cpp:123-141; caller cluster/ClusterFactory.cpp:312) */
	int (*filter)(void* user, uint32_t centre, const uint32_t* points, uint64_t m, uint8_t* keep);
	/* Trainer::merge(centres, current, begin, last) (cluster/Trainer.cpp:74-109; caller cluster/ClusterFactory.cpp:387) */
	int (*merge)(void* user, const uint32_t* centres, uint64_t n, int64_t current, int64_t begin, int64_t last, int64_t* best);
	/* optional (NULL = not available) batched forms of one update round, see msc_update_centres / msc_hist_assign_batch /
	 * msc_merge_all in meshclust2_hip.h */
	int (*update_centres)(void* user, const uint32_t* centres, uint64_t n, const uint32_t* points, const uint64_t* offsets, int64_t* nearest);
	int (*centre_set_batch)(void* user, const uint32_t* centres, const uint32_t* points, uint64_t n);
	int (*merge_all)(void* user, const uint32_t* centres, uint64_t n, int delta, int64_t* best);
} msc_cluster_callbacks;

/* The window of get_close kept on the callee's side (msc_window in meshclust2_hip.h): a struct and an entry point of their own, so that
 * msc_cluster_callbacks keeps the layout clients were compiled against (a struct that grows at its tail is read past its end when an
 * older client passes the shorter one). All three or none.
 *   set_order: order[pos] = point at position pos of the sealed length-binned store; every position starts alive.
 *   get_close_range: get_close over the alive positions of [first, end) in position order; close[0..*n_close) (room for
 *     end - first) = the positions it marks, ascending -- they leave the store with the call; *best = position of the arg-max or -1.
 *   kill: a position that leaves the store otherwise (the next seed: bvec::erase / bvec::pop). */
typedef struct {
	int (*set_order)(void* user, const uint32_t* order, uint64_t n);
	int (*get_close_range)(void* user, uint32_t q, uint64_t first, uint64_t end, uint32_t* close, uint64_t* n_close, int64_t* best, int* is_min);
	int (*kill)(void* user, uint64_t pos);
} msc_cluster_window_callbacks;

/* do_run's tail + ClusterFactory::MS over n points (headers[i] = full header line incl. '>', lengths[i] = effective length).
 * output: .clstr path, or NULL to write nothing (ranks other than 0). log: path of the progress log ("timestamp ..." lines,
 * cluster counts), NULL = stdout. batch_update = 0 takes one centre at a time. err/cap receive the message of a failure.
 * Returns 0, a callback's status, or -1 (err says what). */
int msc_cluster_run(const msc_cluster_callbacks* cb, uint64_t n, const char* const* headers, const uint64_t* lengths, double similarity,
                    int delta, int iterations, const char* output, const char* log, int batch_update, char* err, size_t cap);
/* the same with the window callbacks (wcb may be NULL: then exactly msc_cluster_run); wcb's `user` is cb->user */
int msc_cluster_run_windows(const msc_cluster_callbacks* cb, const msc_cluster_window_callbacks* wcb, uint64_t n, const char* const* headers,
                            const uint64_t* lengths, double similarity, int delta, int iterations, const char* output, const char* log,
                            int batch_update, char* err, size_t cap);

/* ---- the length-binned store on its own (cluster/bvec.{h,cpp}, cluster/bvec_iterator.h), for the CPU fuzz that holds it to the
 * reference's bvec (tests/test_driver_cpu.py). Record i has length lengths[i]; records are added in index order, then sealed. */
void*    msc_bins_create(const uint64_t* lengths, uint64_t n, uint64_t per_bin);
void     msc_bins_destroy(void* bins);
uint64_t msc_bins_count(const void* bins);                                     /* number of bins */
/* records in bin order -> ids_out (as many as are left), sizes_out[bin]; returns the number of records left */
uint64_t msc_bins_layout(const void* bins, uint32_t* ids_out, uint64_t* sizes_out);
/* get_range: out = {front.bin, front.at, back.bin, back.at, back.none} */
void     msc_bins_range(const void* bins, uint64_t begin_len, uint64_t end_len, uint64_t out[5]);
/* the scoring window of [begin_len, end_len] as accumulate walks it (end - begin trips from begin): ids in window order
 * (at most cap written); returns the trip count, which may be negative */
int64_t  msc_bins_window(const void* bins, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out, uint64_t cap);
void     msc_bins_mark(void* bins, uint64_t bin, uint64_t at);
/* remove_available over the range of [begin_len, end_len]: marked records leave, ids in removal order; returns how many */
uint64_t msc_bins_take_marked(void* bins, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out);
int64_t  msc_bins_take_first(void* bins);                                      /* pop: id or -1 */
void     msc_bins_erase(void* bins, uint64_t bin, uint64_t at);

/* ---- Matrix::gaussJordanInverse (predict/Matrix.cpp:109-207) as msc_train_class uses it, for the same fuzz: out = "inverse" of the
 * n x n row-major matrix a (the input itself when the reference would call it singular) */
void     msc_host_inverse(uint64_t n, const double* a, double* out);

#ifdef __cplusplus
}
#endif
#endif /* MESHCLUST2_DRIVER_H */
