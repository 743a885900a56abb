// Device-side build of the RENDER tree: the binned-SAH builder of host/render_bvh.cpp — 32 bins on
// each of the three axes, scale-invariant cost, leaves of at most `max_leaf` primitives, median splits below
// depth 32 — as a breadth-first build on the GPU.  (The reference-topology tree of bvh.rs:131-234, which decides
// equal-t ties and feeds the f64 parity kernel, stays on the host: its stable sorts are inherently sequential.)
//
// Nodes with more than `small` primitives are split level by level, all of a level's nodes at once:
//   bin_kernel        one thread per primitive slot: the slot's primitive goes into its node's 3 x 32 bins
//                     (count, box and centroid box per bin; float min/max as ordered-int atomics);
//   split_kernel      one thread per node of the level: the SAH sweep over the bins of each axis, the children's
//                     boxes and ranges from the bins' unions, the children queued for the next level or for the
//                     small-node pass;
//   partition_kernel  one thread per slot: the primitive moves to its side's range (one atomic per primitive).
// Nodes of `small` primitives or fewer are finished by small_kernel, one thread per node, with the host
// builder's recursion on the node's own range.  Any valid tree gives the same images (closest hit, ties by the
// reference's rank), so the tests compare images and validate the tree's structure, not its shape.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../host/bvh_build.h"

namespace rt {

namespace {

constexpr int kBins = 32;
constexpr int kSmallMax = 64;          // nodes of at most `small` (<= this) primitives are finished by one thread
constexpr int kSmallDefault = 8;    // measured (lucy, teapot): 4..8 best, 16 +0.5 ms, 32 +2 ms, 64 +6 ms
constexpr int kBalancedFromDepth = 32;  // as on the host: median splits from this depth on (root = 1)

struct DNode {
    float lo[3], hi[3];
    int left, right;    // inner: node indices; leaf: -1
    int first, count;   // range in the slot order
    int depth;
    int fill;           // partition cursor of this node's range
    int split_axis;     // -1: not split; 3: split by position in the range
    int split_bin;      // primitives of bins <= split_bin go left
    float clo[3], chi[3];  // centroid bounds (valid for nodes the level passes or the small pass take up)
};

struct Bin {
    unsigned int n;
    int blo[3], bhi[3], clo[3], chi[3];  // ordered-int floats
};

__host__ __device__ inline int ord(float f) {
    const int i = __builtin_bit_cast(int, f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float unord(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }

__device__ inline float area3(const float* lo, const float* hi) {
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    if (!(x >= 0.f && y >= 0.f && z >= 0.f)) return 0.f;
    return 2.f * (x * y + y * z + z * x);
}
__device__ inline int bin_of(float c, float lo, float scale) { return min(kBins - 1, (int)((c - lo) * scale)); }

__global__ void init_bins_kernel(Bin* bins, const int* n_work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_work * 3 * kBins) return;
    Bin b;
    b.n = 0u;
    for (int a = 0; a < 3; ++a) b.blo[a] = b.clo[a] = ord(INFINITY), b.bhi[a] = b.chi[a] = ord(-INFINITY);
    bins[i] = b;
}

// One thread per slot (slots of a node are contiguous).  A block whose 256 slots belong to ONE node of the level
// (every block of the top levels) bins in LDS first and sends each non-empty bin to memory once; mixed blocks
// send every primitive's atomics straight to memory.
__global__ void __launch_bounds__(256) bin_kernel(const float* __restrict__ bounds, const int* __restrict__ ids,
                                                  const int* __restrict__ node_of, const DNode* __restrict__ nodes,
                                                  const int* __restrict__ work_of_node, Bin* bins, int n) {
    __shared__ Bin local[3 * kBins];
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int first_slot = blockIdx.x * blockDim.x, last_slot = min(n, first_slot + (int)blockDim.x) - 1;
    const bool uniform = node_of[first_slot] == node_of[last_slot];  // the same for every thread of the block
    if (uniform) {
        if (threadIdx.x < 3 * kBins) {
            Bin b;
            b.n = 0u;
            for (int a = 0; a < 3; ++a) b.blo[a] = b.clo[a] = ord(INFINITY), b.bhi[a] = b.chi[a] = ord(-INFINITY);
            local[threadIdx.x] = b;
        }
        __syncthreads();
    }
    const int node = slot < n ? node_of[slot] : -1;
    const int w = node >= 0 ? work_of_node[node] : -1;
    if (w >= 0) {
        const DNode& nd = nodes[node];
        const float* b = bounds + 6 * (size_t)ids[slot];
        float c[3];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (b[a] + b[3 + a]);
        for (int axis = 0; axis < 3; ++axis) {
            const float ext = nd.chi[axis] - nd.clo[axis];
            if (!(ext > 0.f)) continue;
            const int k = bin_of(c[axis], nd.clo[axis], (float)kBins / ext);
            Bin* bin = uniform ? &local[axis * kBins + k] : bins + ((size_t)w * 3 + axis) * kBins + k;
            atomicAdd(&bin->n, 1u);
            for (int a = 0; a < 3; ++a) {
                atomicMin(&bin->blo[a], ord(b[a])), atomicMax(&bin->bhi[a], ord(b[3 + a]));
                atomicMin(&bin->clo[a], ord(c[a])), atomicMax(&bin->chi[a], ord(c[a]));
            }
        }
    }
    if (!uniform) return;
    __syncthreads();
    const int wb = work_of_node[node_of[first_slot]];
    if (wb < 0 || threadIdx.x >= 3 * kBins) return;
    const Bin& src = local[threadIdx.x];
    if (!src.n) return;
    Bin* dst = bins + (size_t)wb * 3 * kBins + threadIdx.x;
    atomicAdd(&dst->n, src.n);
    for (int a = 0; a < 3; ++a) {
        atomicMin(&dst->blo[a], src.blo[a]), atomicMax(&dst->bhi[a], src.bhi[a]);
        atomicMin(&dst->clo[a], src.clo[a]), atomicMax(&dst->chi[a], src.chi[a]);
    }
}

struct Side {
    float lo[3], hi[3], clo[3], chi[3];
    unsigned int n;
    __device__ void clear() {
        n = 0;
        for (int a = 0; a < 3; ++a) lo[a] = clo[a] = INFINITY, hi[a] = chi[a] = -INFINITY;
    }
    __device__ void add(const Bin& b) {
        if (!b.n) return;
        n += b.n;
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], unord(b.blo[a])), hi[a] = fmaxf(hi[a], unord(b.bhi[a]));
            clo[a] = fminf(clo[a], unord(b.clo[a])), chi[a] = fmaxf(chi[a], unord(b.chi[a]));
        }
    }
};

__device__ inline void make_child(DNode* nodes, int idx, const Side& s, int first, int depth) {
    DNode c;
    for (int a = 0; a < 3; ++a) c.lo[a] = s.lo[a], c.hi[a] = s.hi[a], c.clo[a] = s.clo[a], c.chi[a] = s.chi[a];
    c.left = c.right = -1, c.first = first, c.count = (int)s.n, c.depth = depth, c.fill = 0, c.split_axis = -1, c.split_bin = 0;
    nodes[idx] = c;
}

// One thread per node of the level.  counters: [0] nodes allocated, [1] next level's work count, [2] small-node
// count, [3] position-split nodes (their children's boxes come from fix_position_kernel)
__global__ void split_kernel(DNode* nodes, const int* __restrict__ work, const int* n_work, const Bin* __restrict__ bins,
                             int* next_work, int* small_list, int* counters, int* work_of_node, int small) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= *n_work) return;
    const int node = work[w];
    DNode nd = nodes[node];
    float best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    const bool balanced = nd.depth >= kBalancedFromDepth;
    if (balanced) {  // the bin boundary of the widest centroid axis that comes closest to the median
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (nd.chi[a] - nd.clo[a] > nd.chi[axis] - nd.clo[axis]) axis = a;
        if (nd.chi[axis] - nd.clo[axis] > 0.f) {
            const Bin* b = bins + ((size_t)w * 3 + axis) * kBins;
            unsigned int c = 0, best_diff = 0xffffffffu;
            for (int k = 0; k + 1 < kBins; ++k) {
                c += b[k].n;
                if (c == 0 || c == (unsigned)nd.count) continue;
                const unsigned int diff = (unsigned)abs((int)c * 2 - nd.count);
                if (diff < best_diff) best_diff = diff, best_axis = axis, best_bin = k;
            }
            // a lopsided "median" (everything but a few in one bin) would not bound the depth: split by position
            if (best_axis >= 0 && best_diff > (unsigned)nd.count / 2u) best_axis = -1;
        }
    } else {
        for (int axis = 0; axis < 3; ++axis) {
            if (!(nd.chi[axis] - nd.clo[axis] > 0.f)) continue;
            const Bin* b = bins + ((size_t)w * 3 + axis) * kBins;
            float right_area[kBins];
            unsigned int right_cnt[kBins];
            Side acc;
            acc.clear();
            for (int k = kBins - 1; k > 0; --k) {
                acc.add(b[k]);
                right_area[k] = area3(acc.lo, acc.hi), right_cnt[k] = acc.n;
            }
            acc.clear();
            for (int k = 0; k + 1 < kBins; ++k) {
                acc.add(b[k]);
                if (acc.n == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = area3(acc.lo, acc.hi) * (float)acc.n + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = k;
            }
        }
    }
    const int kids = atomicAdd(&counters[0], 2);
    Side l, r;
    l.clear(), r.clear();
    if (best_axis >= 0) {
        const Bin* b = bins + ((size_t)w * 3 + best_axis) * kBins;
        for (int k = 0; k < kBins; ++k) (k <= best_bin ? l : r).add(b[k]);
        nd.split_axis = best_axis, nd.split_bin = best_bin;
    } else {  // coincident centroids (or no usable median): halve the range by position; boxes by a pass of their own
        l.n = (unsigned)nd.count / 2u, r.n = (unsigned)nd.count - l.n;
        nd.split_axis = 3, nd.split_bin = 0;
        atomicAdd(&counters[3], 1);
    }
    nd.left = kids, nd.right = kids + 1;
    nodes[node] = nd;
    make_child(nodes, kids, l, nd.first, nd.depth + 1);
    make_child(nodes, kids + 1, r, nd.first + (int)l.n, nd.depth + 1);
    work_of_node[node] = -2;  // partitioned by this level's partition_kernel, then retired
    for (int c = 0; c < 2; ++c) {
        const int idx = kids + c, cnt = (int)(c ? r.n : l.n);
        if (cnt > small) next_work[atomicAdd(&counters[1], 1)] = idx;
        else small_list[atomicAdd(&counters[2], 1)] = idx;
    }
}

// One thread per slot.  A wave's lanes that go to the same child share one cursor atomic (the top levels would
// otherwise send every primitive's atomic to the same two addresses); order inside a child's range is immaterial.
__global__ void __launch_bounds__(256) partition_kernel(const float* __restrict__ bounds, const int* __restrict__ ids_in,
                                                        const int* __restrict__ node_in, int* __restrict__ ids_out,
                                                        int* __restrict__ node_out, DNode* nodes,
                                                        const int* __restrict__ work_of_node, int n) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = slot < n;
    const int node = live ? node_in[slot] : -1, prim = live ? ids_in[slot] : -1;
    const bool splitting = live && work_of_node[node] == -2;
    int child = -1;
    if (splitting) {
        const DNode& nd = nodes[node];
        int side;
        if (nd.split_axis == 3) {
            side = slot - nd.first >= nd.count / 2;
        } else {
            const float* b = bounds + 6 * (size_t)prim;
            const int a = nd.split_axis;
            const float c = 0.5f * (b[a] + b[3 + a]);
            side = bin_of(c, nd.clo[a], (float)kBins / (nd.chi[a] - nd.clo[a])) > nd.split_bin;
        }
        child = nd.left + side;
    } else if (live) {
        ids_out[slot] = prim, node_out[slot] = node;
    }
    unsigned long long todo = __ballot(splitting);
    while (todo) {  // wave-uniform: one trip per distinct child among the wave's lanes
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __shfl(child, leader);
        const bool mine = splitting && child == c;
        const unsigned long long group = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&nodes[c].fill, __popcll(group));
        base = __shfl(base, leader);
        if (mine) {
            const int pos = nodes[c].first + base + __popcll(group & ((1ull << lane) - 1ull));
            ids_out[pos] = prim, node_out[pos] = c;
        }
        todo &= ~group;
    }
}

// children of position-split nodes: their boxes and centroid boxes from their primitives (rare: coincident centroids)
__global__ void fix_position_kernel(const float* __restrict__ bounds, const int* __restrict__ ids, const int* __restrict__ node_of,
                                    const DNode* __restrict__ nodes, int* scratch /* 12 ints per node, pre-set */, int n) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const int node = node_of[slot];
    const DNode& nd = nodes[node];
    if (nd.lo[0] <= nd.hi[0]) return;  // has its box
    int* s = scratch + 12 * (size_t)node;
    const float* b = bounds + 6 * (size_t)ids[slot];
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (b[a] + b[3 + a]);
        atomicMin(&s[a], ord(b[a])), atomicMax(&s[3 + a], ord(b[3 + a]));
        atomicMin(&s[6 + a], ord(c)), atomicMax(&s[9 + a], ord(c));
    }
}
__global__ void apply_position_kernel(DNode* nodes, const int* scratch, int n_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    DNode& nd = nodes[i];
    if (nd.count > 0 && !(nd.lo[0] <= nd.hi[0])) {  // box still empty: a child of a position split
        const int* s = scratch + 12 * (size_t)i;
        for (int a = 0; a < 3; ++a)
            nd.lo[a] = unord(s[a]), nd.hi[a] = unord(s[3 + a]), nd.clo[a] = unord(s[6 + a]), nd.chi[a] = unord(s[9 + a]);
    }
}
// between levels: the nodes just split retire, the next level's nodes learn their place in its work list
__global__ void relabel_kernel(const int* __restrict__ prev, const int* __restrict__ next, const int* counters, int* work_of_node) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < counters[4]) work_of_node[prev[i]] = -1;
    if (i < counters[1]) work_of_node[next[i]] = i;
}
__global__ void advance_kernel(int* counters) {
    counters[4] = counters[1], counters[1] = 0, counters[3] = 0;
}
__global__ void fill_int_kernel(int* p, int v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void reset_position_scratch_kernel(int* s, size_t n_nodes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes * 12) return;
    const int k = (int)(i % 12);
    s[i] = (k < 3 || (k >= 6 && k < 9)) ? ord(INFINITY) : ord(-INFINITY);
}

// ---- nodes of at most `small` primitives: the host builder's recursion, one thread per node, on the node's own range
struct Prim {
    float lo[3], hi[3];
    int id;
};
__device__ inline void load_prim(const float* bounds, int id, Prim& p) {
    const float* b = bounds + 6 * (size_t)id;
    for (int a = 0; a < 3; ++a) p.lo[a] = b[a], p.hi[a] = b[3 + a];
    p.id = id;
}

__global__ void small_kernel(const float* __restrict__ bounds, int* ids, DNode* nodes, const int* __restrict__ small_list,
                             int n_small, int* node_counter, int max_leaf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_small) return;
    struct Task {
        int node, first, n, depth;
    } stack[kSmallMax];
    int sp = 0;
    {
        const DNode& root = nodes[small_list[t]];
        stack[sp++] = Task{small_list[t], root.first, root.count, root.depth};
    }
    while (sp > 0) {
        const Task tk = stack[--sp];
        // box and centroid box of the range
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = tk.first; i < tk.first + tk.n; ++i) {
            const float* b = bounds + 6 * (size_t)ids[i];
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * (b[a] + b[3 + a]);
                lo[a] = fminf(lo[a], b[a]), hi[a] = fmaxf(hi[a], b[3 + a]);
                clo[a] = fminf(clo[a], c), chi[a] = fmaxf(chi[a], c);
            }
        }
        DNode nd = nodes[tk.node];
        for (int a = 0; a < 3; ++a) nd.lo[a] = lo[a], nd.hi[a] = hi[a], nd.clo[a] = clo[a], nd.chi[a] = chi[a];
        nd.first = tk.first, nd.count = tk.n, nd.depth = tk.depth, nd.left = nd.right = -1;
        bool is_leaf = tk.n == 1;
        int mid = tk.first;
        if (!is_leaf && tk.depth >= kBalancedFromDepth) {
            if (tk.n <= max_leaf) {
                is_leaf = true;
            } else {  // median of the widest centroid axis: insertion sort of the (at most kSmallMax) centroids
                int axis = 0;
                for (int a = 1; a < 3; ++a)
                    if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
                for (int i = tk.first + 1; i < tk.first + tk.n; ++i) {
                    const int id = ids[i];
                    const float c = bounds[6 * (size_t)id + axis] + bounds[6 * (size_t)id + 3 + axis];
                    int j = i - 1;
                    while (j >= tk.first && bounds[6 * (size_t)ids[j] + axis] + bounds[6 * (size_t)ids[j] + 3 + axis] > c) {
                        ids[j + 1] = ids[j];
                        --j;
                    }
                    ids[j + 1] = id;
                }
                mid = tk.first + tk.n / 2;
            }
        } else if (!is_leaf) {
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                const float ext = chi[axis] - clo[axis];
                if (!(ext > 0.f)) continue;
                float blo[kBins][3], bhi[kBins][3];
                int cnt[kBins];
                for (int k = 0; k < kBins; ++k) {
                    cnt[k] = 0;
                    for (int a = 0; a < 3; ++a) blo[k][a] = INFINITY, bhi[k][a] = -INFINITY;
                }
                const float scale = (float)kBins / ext;
                for (int i = tk.first; i < tk.first + tk.n; ++i) {
                    const float* b = bounds + 6 * (size_t)ids[i];
                    const int k = bin_of(0.5f * (b[axis] + b[3 + axis]), clo[axis], scale);
                    cnt[k]++;
                    for (int a = 0; a < 3; ++a) blo[k][a] = fminf(blo[k][a], b[a]), bhi[k][a] = fmaxf(bhi[k][a], b[3 + a]);
                }
                float right_area[kBins];
                int right_cnt[kBins];
                float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
                int c = 0;
                for (int k = kBins - 1; k > 0; --k) {
                    for (int a = 0; a < 3; ++a) alo[a] = fminf(alo[a], blo[k][a]), ahi[a] = fmaxf(ahi[a], bhi[k][a]);
                    c += cnt[k];
                    right_area[k] = area3(alo, ahi), right_cnt[k] = c;
                }
                for (int a = 0; a < 3; ++a) alo[a] = INFINITY, ahi[a] = -INFINITY;
                c = 0;
                for (int k = 0; k + 1 < kBins; ++k) {
                    for (int a = 0; a < 3; ++a) alo[a] = fminf(alo[a], blo[k][a]), ahi[a] = fmaxf(ahi[a], bhi[k][a]);
                    c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    const float cost = area3(alo, ahi) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                    if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = k;
                }
            }
            const float parent_area = fmaxf(area3(lo, hi), 1e-30f);
            const float split_cost = 1.0f + best_cost / parent_area;  // traversal cost = intersection cost = 1
            if (tk.n <= max_leaf && (best_axis < 0 || (float)tk.n <= split_cost)) {
                is_leaf = true;
            } else if (best_axis < 0) {
                mid = tk.first + tk.n / 2;
            } else {
                const float scale = (float)kBins / (chi[best_axis] - clo[best_axis]);
                int i = tk.first, j = tk.first + tk.n - 1;
                while (i <= j) {
                    const float* b = bounds + 6 * (size_t)ids[i];
                    if (bin_of(0.5f * (b[best_axis] + b[3 + best_axis]), clo[best_axis], scale) <= best_bin) {
                        ++i;
                    } else {
                        const int tmp = ids[i];
                        ids[i] = ids[j], ids[j] = tmp;
                        --j;
                    }
                }
                mid = i;
                if (mid == tk.first || mid == tk.first + tk.n) mid = tk.first + tk.n / 2;
            }
        }
        if (is_leaf) {
            nodes[tk.node] = nd;
            continue;
        }
        const int kids = atomicAdd(node_counter, 2);
        nd.left = kids, nd.right = kids + 1;
        nodes[tk.node] = nd;
        DNode blank = nd;
        blank.left = blank.right = -1;
        nodes[kids] = blank, nodes[kids + 1] = blank;
        stack[sp++] = Task{kids + 1, mid, tk.first + tk.n - mid, tk.depth + 1};
        stack[sp++] = Task{kids, tk.first, mid - tk.first, tk.depth + 1};
    }
}

#define HIP_TRY(expr)                                                                                                      \
    do {                                                                                                                   \
        hipError_t e_ = (expr);                                                                                            \
        if (e_ != hipSuccess) throw std::runtime_error(std::string("device BVH build: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
    template <class T>
    T* alloc(size_t n) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        ptrs.push_back(p);
        return (T*)p;
    }
    ~DeviceBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

inline unsigned int grid_for(size_t n) { return (unsigned int)((n + 255) / 256); }

}  // namespace

// bounds6: lo.xyz hi.xyz per primitive, f32, rounded outward by the caller; ids[i] = the surface id of primitive i.
void build_render_bvh_device(const std::vector<float>& bounds6, const std::vector<int32_t>& surface_ids, uint32_t max_leaf,
                             const BuildKnobs& knobs, Bvh& out) {
    out = Bvh{};
    const int n = (int)surface_ids.size();
    if (n == 0) return;
    const bool debug = knobs.debug;  // (the environment is read by the caller's thread, never by a worker: host/worker.h)
    const int small = knobs.small ? std::min(kSmallMax, std::max(2, knobs.small)) : kSmallDefault;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = now();
    DeviceBuffers mem;
    const size_t node_cap = 2 * (size_t)n + 2, work_cap = (size_t)n / (size_t)small + 2;
    float* d_bounds = mem.alloc<float>(6 * (size_t)n);
    int* d_ids[2] = {mem.alloc<int>(n), mem.alloc<int>(n)};
    int* d_node_of[2] = {mem.alloc<int>(n), mem.alloc<int>(n)};
    DNode* d_nodes = mem.alloc<DNode>(node_cap);
    int* d_work_of_node = mem.alloc<int>(node_cap);
    int* d_work[2] = {mem.alloc<int>(work_cap), mem.alloc<int>(work_cap)};
    int* d_small = mem.alloc<int>((size_t)n + 2);
    int* d_counters = mem.alloc<int>(8);  // [0] nodes [1] next work [2] small [3] position splits [4] this level's work
    Bin* d_bins = mem.alloc<Bin>(work_cap * 3 * kBins);
    int* d_pos_scratch = nullptr;

    HIP_TRY(hipMemcpy(d_bounds, bounds6.data(), 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    {
        std::vector<int> iota(n);
        for (int i = 0; i < n; ++i) iota[i] = i;
        HIP_TRY(hipMemcpy(d_ids[0], iota.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(d_node_of[0], 0, (size_t)n * sizeof(int)));
    hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(node_cap)), dim3(256), 0, nullptr, d_work_of_node, -1, node_cap);

    // the root: box and centroid box on the host (one pass over the bounds)
    DNode root{};
    for (int a = 0; a < 3; ++a) root.lo[a] = root.clo[a] = INFINITY, root.hi[a] = root.chi[a] = -INFINITY;
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float lo = bounds6[6 * (size_t)i + a], hi = bounds6[6 * (size_t)i + 3 + a], c = 0.5f * (lo + hi);
            root.lo[a] = std::fmin(root.lo[a], lo), root.hi[a] = std::fmax(root.hi[a], hi);
            root.clo[a] = std::fmin(root.clo[a], c), root.chi[a] = std::fmax(root.chi[a], c);
        }
    root.left = root.right = -1, root.first = 0, root.count = n, root.depth = 1, root.fill = 0, root.split_axis = -1;
    HIP_TRY(hipMemcpy(d_nodes, &root, sizeof root, hipMemcpyHostToDevice));
    int counters[8] = {1, 0, 0, 0, 0, 1, 0, 0};
    int cur = 0, n_work = 0;
    if (n > small) {
        const int zero = 0;
        HIP_TRY(hipMemcpy(d_work[0], &zero, sizeof zero, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_work_of_node, &zero, sizeof zero, hipMemcpyHostToDevice));
        n_work = 1;
    } else {
        const int zero = 0;
        HIP_TRY(hipMemcpy(d_small, &zero, sizeof zero, hipMemcpyHostToDevice));
        counters[2] = 1;
    }
    counters[4] = n_work;
    HIP_TRY(hipMemcpy(d_counters, counters, sizeof counters, hipMemcpyHostToDevice));

    const auto t_levels = now();
    int level = 0, w = 0;  // w: which work list is this level's
    while (n_work > 0) {
        if (++level > 60) throw std::runtime_error("device BVH build: too many levels");
        hipLaunchKernelGGL(init_bins_kernel, dim3(grid_for((size_t)n_work * 3 * kBins)), dim3(256), 0, nullptr, d_bins, d_counters + 4);
        hipLaunchKernelGGL(bin_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, d_bounds, d_ids[cur], d_node_of[cur], d_nodes,
                           d_work_of_node, d_bins, n);
        hipLaunchKernelGGL(split_kernel, dim3(grid_for(n_work)), dim3(256), 0, nullptr, d_nodes, d_work[w], d_counters + 4, d_bins,
                           d_work[w ^ 1], d_small, d_counters, d_work_of_node, small);
        hipLaunchKernelGGL(partition_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, d_bounds, d_ids[cur], d_node_of[cur],
                           d_ids[cur ^ 1], d_node_of[cur ^ 1], d_nodes, d_work_of_node, n);
        cur ^= 1;
        HIP_TRY(hipMemcpy(counters, d_counters, sizeof counters, hipMemcpyDeviceToHost));  // synchronises
        if (counters[3] > 0) {
            // children of position splits got no boxes from bins: one pass over the slots fills them in
            if (!d_pos_scratch) d_pos_scratch = mem.alloc<int>(12 * node_cap);
            hipLaunchKernelGGL(reset_position_scratch_kernel, dim3(grid_for(12 * node_cap)), dim3(256), 0, nullptr, d_pos_scratch, node_cap);
            hipLaunchKernelGGL(fix_position_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, d_bounds, d_ids[cur], d_node_of[cur],
                               d_nodes, d_pos_scratch, n);
            hipLaunchKernelGGL(apply_position_kernel, dim3(grid_for(counters[0])), dim3(256), 0, nullptr, d_nodes, d_pos_scratch, counters[0]);
        }
        // the next level: its work list is the one split_kernel appended to
        const int n_next = counters[1];
        hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(std::max(n_work, n_next))), dim3(256), 0, nullptr, d_work[w], d_work[w ^ 1],
                           d_counters, d_work_of_node);
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, nullptr, d_counters);
        n_work = n_next, w ^= 1;
    }
    const auto t_small = now();
    // the small nodes, one thread each
    if (counters[2] > 0) {
        hipLaunchKernelGGL(small_kernel, dim3((unsigned)((counters[2] + 63) / 64)), dim3(64), 0, nullptr, d_bounds, d_ids[cur], d_nodes,
                           d_small, counters[2], d_counters, (int)std::max(1u, max_leaf));
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(counters, d_counters, sizeof counters, hipMemcpyDeviceToHost));
    const auto t_down = now();
    const int n_nodes = counters[0];
    if ((size_t)n_nodes > node_cap) throw std::runtime_error("device BVH build: node pool overflow");
    std::vector<DNode> nodes(n_nodes);
    std::vector<int> order(n);
    HIP_TRY(hipMemcpy(nodes.data(), d_nodes, (size_t)n_nodes * sizeof(DNode), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(order.data(), d_ids[cur], (size_t)n * sizeof(int), hipMemcpyDeviceToHost));

    out.nodes.resize(n_nodes);
    for (int i = 0; i < n_nodes; ++i) {
        const DNode& d = nodes[i];
        BvhNode& b = out.nodes[i];
        b.box = Aabb{{d.lo[0], d.lo[1], d.lo[2]}, {d.hi[0], d.hi[1], d.hi[2]}};
        b.left = d.left, b.right = d.right;
        b.first = d.first, b.count = d.count;
        if (b.is_leaf()) {
            out.n_leaves++;
            out.max_leaf_size = std::max(out.max_leaf_size, (uint32_t)d.count);
            out.max_depth = std::max(out.max_depth, (uint32_t)d.depth);
        }
    }
    out.order.resize(n);
    for (int i = 0; i < n; ++i) out.order[i] = surface_ids[order[i]];
    if (debug)
        fprintf(stderr, "device BVH build: %d primitives, %d levels, %d small nodes, %d nodes; alloc+upload %.2f ms, levels %.2f ms, "
                        "small nodes %.2f ms, download+convert %.2f ms\n",
                n, level, counters[2], n_nodes, ms(t_start, t_levels), ms(t_levels, t_small), ms(t_small, t_down), ms(t_down, now()));
}

}  // namespace rt
