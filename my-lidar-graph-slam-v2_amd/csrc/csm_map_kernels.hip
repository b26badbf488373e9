/* csm_map_kernels.hip -- device side of the map updates
 * (GridMapBuilder::ConstructMapFromScans / UpdateGridMap,
 * src/my_lidar_graph_slam/mapping/grid_map_builder.cpp:389-494, 561-695).
 * Included by csm_map_api.hip (its own translation unit); cell_index and
 * proj_err_bound come from csm_score_common.hpp. gfx950 only. */
#ifndef CSM_MAP_KERNELS_HIP
#define CSM_MAP_KERNELS_HIP

#include "csm_score_common.hpp"

namespace csm {


/* GridMapBuilder::ConstructMapFromScans (src/mapping/grid_map_builder.cpp:647-692)
 * applies, ray after ray, a miss update to every cell the ray crosses and a hit
 * update to its end cell. A cell update is a function value -> value, so what a
 * cell ends as depends only on ITS sequence of hits and misses in ray order.
 * A cell no ray ends in needs just its miss count; for the others the hits are
 * ranked by ray number and the misses are counted per interval between
 * consecutive hits. Six small kernels, integer atomics only, no ordering
 * assumed between threads. */

/* ScanData::HitPoint for every beam (grid_map_builder.cpp:614-630) with the
 * device's sin / cos. Only integers derived from the hit point are used later --
 * floor((h - off) / res) at the cell and the sub-pixel resolution, in the frame
 * the resize will choose, and floor((h -+ res - off) / res) for the bounding box
 * -- and all of them are q + (a whole number) up to roundings of ~1e-12 cells,
 * where q is the same expression in the CURRENT frame. A beam whose q (cell and
 * sub-pixel scale, both axes) stays farther from the next integer than the two
 * libms can disagree plus that slack gives the host's integers; the others are
 * listed and recomputed on the host with glibc. */
__device__ __forceinline__ bool map_certified(double r, double h, double off, double res)
{
    const double q = (h - off) / res;
    const double m = 64.0 * proj_err_bound(r, h, off, res, q, 8e-16) +    /* 3 ulp between the two libms */
                     64.0 * 2.3e-16 * (fabs(h) + fabs(off) + 1.0e3) / res;
    const double frac = q - floor(q);
    return frac > m && frac < 1.0 - m;
}

__global__ __launch_bounds__(256) void k_map_project(MapProjJob job)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    int lo_x = 0x7fffffff, lo_y = 0x7fffffff, hi_x = -0x7fffffff - 1, hi_y = -0x7fffffff - 1;
    if (b < job.n_beams) {
        /* the node of this beam: nodes are few, beams ordered by node */
        int k = 0;
        while (k + 1 < job.n_nodes && job.nodes[k + 1].beam_base <= b)
            ++k;
        const MapNode nd = job.nodes[k];
        const double r = job.ranges[b];
        MapRay ray = { 0.0, 0.0, k, 0 };
        if (!(r >= nd.max_range || r <= nd.min_range)) {
            const double arg = nd.theta + job.angles[b];
            ray.hx = nd.x + r * cos(arg);
            ray.hy = nd.y + r * sin(arg);
            ray.usable = 1;
            /* clearly off the sensor position in x / y: the bounding box is not degenerate */
            const uint32_t spread = (fabs(ray.hx - nd.x) > 1e-6 ? 1u : 0u) | (fabs(ray.hy - nd.y) > 1e-6 ? 2u : 0u);
            if (spread & ~job.unc_count[1])
                atomicOr(&job.unc_count[1], spread);
            const bool sure = map_certified(r, ray.hx, job.off_x, job.res) &&
                              map_certified(r, ray.hy, job.off_y, job.res) &&
                              map_certified(r, ray.hx, job.off_x, job.scaled_res) &&
                              map_certified(r, ray.hy, job.off_y, job.scaled_res);
            if (sure) {
                /* GridMap::Resize(BoundingBox<double>) (grid_map.cpp:892-913) is monotone in h */
                lo_x = cell_index(ray.hx - job.res, job.off_x, job.res);
                lo_y = cell_index(ray.hy - job.res, job.off_y, job.res);
                hi_x = cell_index(ray.hx + job.res, job.off_x, job.res);
                hi_y = cell_index(ray.hy + job.res, job.off_y, job.res);
            } else {
                const uint32_t pos = atomicAdd(job.unc_count, 1u);
                if (pos < job.unc_cap)
                    job.unc_list[pos] = (uint32_t)b;
            }
        }
        job.rays[b] = ray;
    }
    for (int off = 32; off; off >>= 1) {
        lo_x = min(lo_x, __shfl_xor(lo_x, off));
        lo_y = min(lo_y, __shfl_xor(lo_y, off));
        hi_x = max(hi_x, __shfl_xor(hi_x, off));
        hi_y = max(hi_y, __shfl_xor(hi_y, off));
    }
    if ((threadIdx.x & 63) == 0 && lo_x != 0x7fffffff) {
        atomicMin(&job.box[0], lo_x);
        atomicMin(&job.box[1], lo_y);
        atomicMax(&job.box[2], hi_x);
        atomicMax(&job.box[3], hi_y);
    }
}

/* hit cell + sub-pixel end of every ray; the cell's hit counter hands out slots */
__global__ __launch_bounds__(256) void k_map_hits(MapJob job)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= job.n_rays)
        return;
    const MapRay ray = job.rays[r];
    MapRayRec rec = { 0, 0, -1, 0 };
    if (ray.usable) {
        const int sx = job.nodes[ray.node].sx, sy = job.nodes[ray.node].sy;
        const int col = cell_index(ray.hx, job.off_x, job.res);
        const int row = cell_index(ray.hy, job.off_y, job.res);
        rec.ex = cell_index(ray.hx, job.off_x, job.scaled_res);
        rec.ey = cell_index(ray.hy, job.off_y, job.scaled_res);
        const bool ok = col >= 0 && col < job.cols && row >= 0 && row < job.rows && sx >= 0 && sy >= 0 &&
                        rec.ex >= 0 && rec.ey >= 0 && rec.ex / job.scale < job.cols &&
                        rec.ey / job.scale < job.rows && sx / job.scale < job.cols &&
                        sy / job.scale < job.rows;
        if (ok) {
            rec.hit_cell = row * job.cols + col;
            rec.slot = (int)atomicAdd(&job.n_hit[rec.hit_cell], 1u);
        } else {
            atomicOr(&job.counters[kMapError], 1ull);   /* the reference asserts (bresenham.cpp:73-76) */
        }
    }
    job.recs[r] = rec;
}

__global__ __launch_bounds__(256) void k_map_alloc(MapJob job)
{
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t n = cell < job.rows * job.cols ? job.n_hit[cell] : 0u;
    /* one pair of atomics per wavefront: prefix sums of the block sizes and of the hit cells */
    uint32_t words = n ? map_block_words(n) : 0u, cells = n ? 1u : 0u;
    uint32_t words_incl = words, cells_incl = cells;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t w = __shfl_up(words_incl, off), c = __shfl_up(cells_incl, off);
        if (lane >= off) {
            words_incl += w;
            cells_incl += c;
        }
    }
    const uint32_t words_total = __shfl(words_incl, 63), cells_total = __shfl(cells_incl, 63);
    if (cells_total == 0)
        return;
    uint32_t words_base = 0, cells_base = 0;
    if (lane == 0) {
        words_base = (uint32_t)atomicAdd(&job.counters[kMapCursor], (unsigned long long)words_total);
        cells_base = (uint32_t)atomicAdd(&job.counters[kMapHitCells], (unsigned long long)cells_total);
    }
    words_base = __shfl(words_base, 0);
    cells_base = __shfl(cells_base, 0);
    if (n) {
        const uint32_t base = words_base + words_incl - words;
        job.seg[cell] = base;
        job.hit_cells[cells_base + cells_incl - 1] = (uint32_t)cell;
        uint4* between = reinterpret_cast<uint4*>(job.lists + base + map_between_offset(n));
        for (uint32_t i = 0; i < (n + 4u) / 4u; ++i)
            between[i] = make_uint4(0, 0, 0, 0);
    }
}

__global__ __launch_bounds__(256) void k_map_fill_hits(MapJob job)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= job.n_rays)
        return;
    const MapRayRec rec = job.recs[r];
    if (rec.hit_cell >= 0)
        job.lists[job.seg[rec.hit_cell] + rec.slot] = (uint32_t)r;
}

/* rank of each hit among its cell's hits = its place in ray order */
__global__ __launch_bounds__(256) void k_map_rank_hits(MapJob job)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= job.n_rays)
        return;
    const MapRayRec rec = job.recs[r];
    if (rec.hit_cell < 0)
        return;
    const uint32_t n = job.n_hit[rec.hit_cell];
    const uint32_t* arrival = job.lists + job.seg[rec.hit_cell];
    uint32_t rank = 0;
    for (uint32_t i = 0; i < n; ++i)
        rank += arrival[i] < (uint32_t)r;
    job.lists[job.seg[rec.hit_cell] + n + rank] = (uint32_t)r;
}

/* the rays of one workgroup of k_map_walk count the misses of hit-free cells
 * inside this window in LDS first */
struct MapWindow {
    int x_lo, y_lo, w, h;
    uint32_t* count;
};

__device__ __forceinline__ void map_miss(const MapJob& job, const MapWindow& win, int x, int y,
                                         int skip_x, int skip_y, uint32_t r)
{
    if (x == skip_x && y == skip_y)
        return;                              /* the end cell is taken off the list (grid_map_builder.cpp:904-910) */
    if (x < 0 || x >= job.cols || y < 0 || y >= job.rows) {
        atomicOr(&job.counters[kMapError], 2ull);
        return;
    }
    const int cell = y * job.cols + x;
    const uint32_t n = job.n_hit[cell];
    if (n == 0) {
        const int wx = x - win.x_lo, wy = y - win.y_lo;
        if ((unsigned)wx < (unsigned)win.w && (unsigned)wy < (unsigned)win.h)
            atomicAdd(&win.count[wy * win.w + wx], 1u);
        else
            atomicAdd(&job.n_miss[cell], 1u);
        return;
    }
    /* number of this cell's hits that come before ray r (a ray's own hit comes after its misses) */
    const uint32_t base = job.seg[cell];
    const uint32_t* sorted = job.lists + base + n;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] < r)
            lo = mid + 1;
        else
            hi = mid;
    }
    atomicAdd(&job.lists[base + map_between_offset(n) + lo], 1u);
}

/* One wavefront per ray, lanes over the ray's cell columns. The cells are those
 * of BresenhamScaled (src/bresenham.cpp:58-237) in closed form: with the ray's
 * height N(x) counted in 1/(2 * scale * dx) cells, column j holds the rows from
 * where the ray enters it to where it leaves it; a ray that leaves through an
 * exact cell corner steps diagonally (the corner's other two cells are not
 * visited). */
__device__ __forceinline__ void map_walk_ray(const MapJob& job, const MapWindow& win, int r, int lane)
{
    const MapRayRec rec = job.recs[r];
    if (rec.hit_cell < 0)
        return;
    const int scale = job.scale;
    const MapNode& node = job.nodes[job.rays[r].node];
    int sx = node.sx, sy = node.sy, ex = rec.ex, ey = rec.ey;
    const int skip_x = ex / scale, skip_y = ey / scale;
    if (sx > ex) {                            /* bresenham.cpp:67-70 */
        int t = sx; sx = ex; ex = t;
        t = sy; sy = ey; ey = t;
    }
    const int x0 = sx / scale, y0 = sy / scale, x1 = ex / scale, y1 = ey / scale;
    if (x0 == x1) {                           /* bresenham.cpp:87-99 */
        const int lo = min(y0, y1), hi = max(y0, y1);
        for (int y = lo + lane; y <= hi; y += 64)
            map_miss(job, win, x0, y, skip_x, skip_y, (uint32_t)r);
        return;
    }
    const long long dx = ex - sx, dy = ey - sy;
    const long long den = 2ll * scale * dx;
    const long long n0 = (long long)y0 * den + (2ll * (sy % scale) + 1) * dx;
    const long long first = 2ll * scale - (2ll * (sx % scale) + 1);
    const long long last = 2ll * (ex % scale) + 1;
    const int m = x1 - x0;
    for (int j0 = 0; j0 <= m; j0 += 64) {
        const int j = j0 + lane;
        int from = 0, count = 0;
        if (j <= m) {
            const long long n_out = j < m ? n0 + dy * (first + 2ll * scale * j)
                                          : n0 + dy * (first + 2ll * scale * (m - 1) + last);
            const long long n_in = n0 + dy * (first + 2ll * scale * (j - 1));   /* unused for j = 0 */
            int to;
            if (dy > 0) {
                from = j == 0 ? y0 : (int)(n_in / den);
                to = (int)((n_out + den - 1) / den) - 1;
            } else {
                to = j == 0 ? y0 : (int)((n_in + den - 1) / den) - 1;
                from = (int)(n_out / den);
            }
            count = to - from + 1;
        }
        /* steep rays have few columns with many rows each: spread the cells of
         * these 64 columns evenly over the lanes (prefix sum + search) */
        int incl = count;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off)
                incl += up;
        }
        const int excl = incl - count;
        const int total = __shfl(incl, 63);
        for (int t0 = 0; t0 < total; t0 += 64) {
            const int t = t0 + lane;
            int c = 0;                        /* last column whose first cell number is <= t */
            for (int step = 32; step; step >>= 1) {
                const int e = __shfl(excl, min(c + step, 63));
                if (c + step < 64 && e <= t)
                    c += step;
            }
            const int y = __shfl(from, c) + (t - __shfl(excl, c));
            if (t < total)
                map_miss(job, win, x0 + j0 + c, y, skip_x, skip_y, (uint32_t)r);
        }
    }
}

/* kMapGroup consecutive rays per workgroup (neighbouring beams of one scan: they
 * cross the same cells near the sensor). Device-scope atomics execute at the
 * memory side and are the rate limit of this step, so the group first counts
 * in an LDS window over its rays' bounding box and then adds each non-zero
 * counter once, row-contiguous. */
constexpr int kMapGroup = 32;
constexpr int kMapWindowCells = 12288;       /* 48 KB */

__global__ __launch_bounds__(512) void k_map_walk(MapJob job)
{
    __shared__ int box[6];                   /* x_lo, y_lo, x_hi, y_hi, anchor x, anchor y */
    __shared__ uint32_t window[kMapWindowCells];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * kMapGroup;
    if (tid == 0) {
        box[0] = box[1] = 0x7fffffff;
        box[2] = box[3] = box[4] = box[5] = -1;
    }
    __syncthreads();
    if (tid < kMapGroup && r0 + tid < job.n_rays) {
        const MapRayRec rec = job.recs[r0 + tid];
        if (rec.hit_cell >= 0) {
            const MapNode& node = job.nodes[job.rays[r0 + tid].node];
            const int ax = node.sx / job.scale, ay = node.sy / job.scale;
            const int bx = rec.ex / job.scale, by = rec.ey / job.scale;
            atomicMin(&box[0], min(ax, bx));
            atomicMin(&box[1], min(ay, by));
            atomicMax(&box[2], max(ax, bx));
            atomicMax(&box[3], max(ay, by));
            atomicMax(&box[4], ax);          /* any one sensor cell of the group */
            atomicMax(&box[5], ay);
        }
    }
    __syncthreads();
    MapWindow win = { box[0], box[1], box[2] - box[0] + 1, box[3] - box[1] + 1, window };
    if (box[2] < 0)
        return;                              /* no usable ray in this group (uniform) */
    if ((long long)win.w * win.h > kMapWindowCells) {
        /* too large: keep the part around the sensor, where the rays overlap most */
        const double f = 0.95 * sqrt((double)kMapWindowCells / ((double)win.w * win.h));
        const int ax = min(max(box[4], box[0]), box[2]), ay = min(max(box[5], box[1]), box[3]);
        const int x_lo = ax - (int)((ax - box[0]) * f), x_hi = ax + (int)((box[2] - ax) * f);
        const int y_lo = ay - (int)((ay - box[1]) * f), y_hi = ay + (int)((box[3] - ay) * f);
        win.x_lo = x_lo;
        win.y_lo = y_lo;
        win.w = min(x_hi - x_lo + 1, kMapWindowCells);
        win.h = min(y_hi - y_lo + 1, kMapWindowCells / win.w);
    }
    const int cells = win.w * win.h;
    for (int i = tid; i < cells; i += 512)
        window[i] = 0;
    __syncthreads();
    for (int k = wave; k < kMapGroup; k += 8)
        if (r0 + k < job.n_rays)
            map_walk_ray(job, win, r0 + k, lane);
    __syncthreads();
    for (int i = tid; i < cells; i += 512) {
        const uint32_t c = window[i];
        if (c) {
            const int wy = i / win.w, wx = i - wy * win.w;
            atomicAdd(&job.n_miss[(win.y_lo + wy) * job.cols + win.x_lo + wx], c);
        }
    }
}

/* k updates of one kind; stops at a fixed point. Reads of table entry 65535 are
 * counted: the reference's odds table ends at 65534 (grid_values.cpp:74-77). */
__device__ __forceinline__ uint32_t map_iterate(const uint16_t* lut, uint32_t v, uint32_t k, uint32_t& sat)
{
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t nv = lut[v];
        if (nv == v) {
            if (v == 65535u)
                sat += k - i;
            break;
        }
        sat += v == 65535u;
        v = nv;
    }
    return v;
}

/* per-wave totals of k_map_apply / k_map_apply_hits, one atomic each */
__device__ __forceinline__ void map_apply_totals(const MapJob& job, uint32_t v, int row, int col,
                                                 uint32_t sat, uint32_t updates)
{
    uint32_t krow = v ? (uint32_t)row : 0xffffffffu, kcol = v ? (uint32_t)col : 0xffffffffu;
    for (int off = 32; off; off >>= 1) {
        sat += __shfl_xor(sat, off);
        updates += __shfl_xor(updates, off);
        krow = min(krow, (uint32_t)__shfl_xor(krow, off));
        kcol = min(kcol, (uint32_t)__shfl_xor(kcol, off));
    }
    if ((threadIdx.x & 63) == 0) {
        const int stripe = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (kMapStripes - 1);
        if (sat)
            atomicAdd(&job.counters[kMapStripedSaturated + stripe], (unsigned long long)sat);
        if (updates)
            atomicAdd(&job.counters[kMapStripedUpdates + stripe], (unsigned long long)updates);
        /* the minima only ever fall: skip the atomic when a (possibly stale) read already beats us */
        if (krow != 0xffffffffu && (unsigned long long)krow < job.counters[kMapKnownRow])
            atomicMin(&job.counters[kMapKnownRow], (unsigned long long)krow);
        if (kcol != 0xffffffffu && (unsigned long long)kcol < job.counters[kMapKnownCol])
            atomicMin(&job.counters[kMapKnownCol], (unsigned long long)kcol);
    }
}

/* cells no ray ends in: their miss count through the miss table; clears the rest */
__global__ __launch_bounds__(256) void k_map_apply(MapJob job)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid / job.pitch, col = gid - row * job.pitch;
    uint32_t v = 0, sat = 0, updates = 0;
    bool write = row < job.rows;
    if (row < job.rows && col < job.cols) {
        const int cell = row * job.cols + col;
        if (job.n_hit[cell] == 0) {
            updates = job.n_miss[cell];
            const uint32_t start = job.keep_cells ? job.cells[(size_t)row * job.pitch + col] : 0u;
            v = map_iterate(job.lut_miss, start, updates, sat);
        } else {
            write = false;                   /* k_map_apply_hits owns this cell */
        }
    }
    if (write)
        job.cells[(size_t)row * job.pitch + col] = (uint16_t)v;
    map_apply_totals(job, v, row, col, sat, updates);
}

/* Cells with hits: a chain of dependent table reads per cell (hit, a few
 * misses, hit, ...), as long as the cell has hits and misses (a wall cell seen
 * from 10 scans: ~100 + ~100). The longest chain's latency is the kernel's
 * duration. So: the hit table sits in LDS (128 KB per workgroup); the miss
 * counts are fetched four intervals at a time, one fetch ahead; and as few
 * lanes of a wavefront as the cell count allows carry a cell, because a table
 * gather costs per distinct cache line (64 cells per wave: ~340 ns per step,
 * 4: ~100, 1: ~70). */
__global__ __launch_bounds__(256) void k_map_apply_hits(MapJob job)
{
    extern __shared__ uint16_t hit_table[];
    const uint32_t n_cells = (uint32_t)job.counters[kMapHitCells];
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t per_wave = min(max((n_cells + waves - 1u) / waves, 1u), 64u);
    if (blockIdx.x * 4u * per_wave >= n_cells)
        return;                              /* fewer cells than workgroups (uniform exit) */
    {
        const uint4* src = reinterpret_cast<const uint4*>(job.lut_hit);
        uint4* dst = reinterpret_cast<uint4*>(hit_table);
        for (int i = threadIdx.x; i < 65536 * 2 / 16; i += 256)
            dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    for (uint32_t first = 0; first < n_cells; first += waves * per_wave) {
        const uint32_t idx = first + wave * per_wave + lane;
        uint32_t v = 0, sat = 0, updates = 0;
        int row = 0, col = 0;
        if (lane < per_wave && idx < n_cells) {
            const int cell = (int)job.hit_cells[idx];
            row = cell / job.cols;
            col = cell - row * job.cols;
            const uint32_t n = job.n_hit[cell];
            if (job.keep_cells)
                v = job.cells[(size_t)row * job.pitch + col];
            const uint4* between = reinterpret_cast<const uint4*>(job.lists + job.seg[cell] + map_between_offset(n));
            uint4 cur = between[0];
            for (uint32_t i0 = 0; i0 <= n; i0 += 4) {
                const uint4 nxt = i0 + 4 <= n ? between[i0 / 4 + 1] : make_uint4(0, 0, 0, 0);
                const uint32_t k4[4] = { cur.x, cur.y, cur.z, cur.w };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t i = i0 + j;
                    if (i <= n) {
                        updates += k4[j];
                        v = map_iterate(job.lut_miss, v, k4[j], sat);
                        if (i < n) {
                            sat += v == 65535u;
                            v = hit_table[v];
                        }
                    }
                }
                cur = nxt;
            }
            updates += n;
            job.cells[(size_t)row * job.pitch + col] = (uint16_t)v;
        }
        map_apply_totals(job, v, row, col, sat, updates);
    }
}

} /* namespace csm */
#endif
