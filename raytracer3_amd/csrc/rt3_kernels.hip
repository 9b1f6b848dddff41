// rt3_kernels.hip -- gfx950 wavefront path-tracing kernels (hand-written HIP, wave64).
//
// Pipeline (replaces shaders/old/{gbuffer,refrence_mode,postprocess}.slang + the driver's ray traversal):
//   k_raygen -> k_extend -> k_gbuffer                                   ("gbuffer" pass)
//   k_shade<first> -> [k_shadow] -> k_extend -> k_shade -> ... -> k_accumulate   ("refrence_mode" pass; k_trace = k_shadow + k_extend in one launch)
//   k_postprocess                                                        ("postprocess" pass)
// All queues are structure-of-arrays of 16-byte records (ray = {o, tmin} + {d, tmax}, state = {T, pdf}, hit = {t, u, v, prim};
// shadow ray = {o, contribution.r} + {d, contribution.g} + 8 bytes {contribution.b, path id}): lane i touches record i of each
// stream, 1 KiB per wave instruction, the widest coalesced access; live rays are compacted with __ballot / popcount, one
// 64-bit atomic per workgroup for both output queues.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "rt3_device.hpp"
#include "rt3_internal.hpp"

namespace rt3 {

// ------------------------------------------------------------------------------------------------ traversal
// One ray per lane.  Short stack: kLdsStack entries per lane in LDS ([entry][lane] so a wave's ds_read_b32 /
// ds_write_b32 hit 64 consecutive dwords: conflict-free), deeper entries spill to a private array (scratch).
// LDS per 256-thread block: 12 KiB of stack + 8 KiB of cached top-of-tree nodes = 20 KiB -> 8 blocks = 8 waves per SIMD in 160 KiB.
// Stack: a diffuse ray's stack is 4 entries deep at the median, 6 at the 90th and 9 at the 99th percentile (13 at most on the atrium).
constexpr int kLdsStack = 12;
// Top-of-tree cache (quantised four-wide layout): the first kTopNodes nodes of the tree in breadth-first order live in LDS, copied at
// kernel start from a 8 KiB array the builder prepares (k_top_cache).  They receive 46 % of all node visits (the root alone 5 %), and the
// walk is bound by the rate at which the vector-memory path returns gathered bytes (profiles/r02_gather_cap.md): these visits now go
// through the LDS instead.  In the cached copies a reference to a child that is itself cached is kTopFlag | slot.
constexpr uint32_t kTopFlag = 0x40000000u;
constexpr int kSpill = 64 - kLdsStack;  // kLdsStack + kSpill >= kMaxBvhDepth (checked on the host after the build)
constexpr uint32_t kMaxSteps = 1u << 20;  // safety bound on traversal steps per ray (a corrupt tree must not hang the GPU)

constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;

struct Cand {  // a child slot that the ray enters: entry distance + reference
    float tn;
    uint32_t ref;
};
// compare-exchange on the entry distance alone (strict <).  Equal distances are ordered by the fixed 5-comparator
// network itself; the oracle runs the identical network, so the visiting order is the same on both sides.
__device__ __forceinline__ void cswap(Cand& a, Cand& b) {
    bool sw = b.tn < a.tn;
    float ta = a.tn;
    uint32_t ra = a.ref;
    a.tn = sw ? b.tn : a.tn;
    a.ref = sw ? b.ref : a.ref;
    b.tn = sw ? ta : b.tn;
    b.ref = sw ? ra : b.ref;
}
__device__ __forceinline__ void pin(float4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }
// Watertight ray / triangle test, branch-free (the oracle's tri_test_dd has the argument): edge functions as signed volumes
// U = d.(B x C), V = d.(C x A), W = d.(A x B) of the vertices relative to the ray origin, every cross-product component two rounded
// products and a subtraction (NO fma: the file is compiled with -ffp-contract=off), so the two triangles of a shared edge compute
// the same number for it up to sign and no ray passes between them.  No early-outs: the three loads of a triangle are issued
// together instead of being sunk behind branches.  Record: {v0.xyz, v1.x} {v1.yz, v2.xy} {v2.z, prim, -, -}.
__device__ __forceinline__ V3 cross_exact(V3 a, V3 b) { return V3{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
__device__ __forceinline__ void tri_test_nb(float4 q0, float4 q1, float4 q2, V3 o, V3 d, float inv_dd, float tmin, Hit& best) {
    const V3 A = v3(q0.x, q0.y, q0.z) - o, B = v3(q0.w, q1.x, q1.y) - o, C = v3(q1.z, q1.w, q2.x) - o;
    const float U = dot_fma(d, cross_exact(B, C)), V = dot_fma(d, cross_exact(C, A)), W = dot_fma(d, cross_exact(A, B));
    const float det = U + (V + W);  // the association of T below: equal vertex distances give t exactly
    const float inv = 1.0f / det;
    const float w = U * inv, u = V * inv, v = W * inv;
    const float T = __builtin_fmaf(U, dot_fma(A, d), __builtin_fmaf(V, dot_fma(B, d), W * dot_fma(C, d)));
    const float t = (T * inv) * inv_dd;
    const uint32_t prim = __float_as_uint(q2.y);
    // barycentrics >= -2^-20: a superset of "U, V, W share a sign" (the shared-edge guarantee stands) that also closes T-junctions
    // and edges of separate meshes that merely coincide, which no watertight test covers
    constexpr float kEdgeEps = 9.5367431640625e-07f;
    const bool inside = (w >= -kEdgeEps) & (u >= -kEdgeEps) & (v >= -kEdgeEps);
    const bool ok = (det != 0.0f) & inside & (t > tmin) & ((t < best.t) | ((t == best.t) & (prim < best.prim)));
    best.t = ok ? t : best.t;
    best.u = ok ? u : best.u;
    best.v = ok ? v : best.v;
    best.prim = ok ? prim : best.prim;
}

// Persistent-wave traversal.  The waves of a launch share one pool of rays (chunks of g_pool_chunk, handed out by an atomic
// cursor).  One ray per lane; a lane that finishes its ray takes the next one from the wave's current chunk (refill once
// >= g_refill_lanes lanes are idle), so the wave keeps its 64 lanes busy instead of idling until its slowest ray is done: a
// plain one-ray-per-lane loop spent ~49 iterations per 64 rays whose mean length is ~23 steps.
//
// Layouts: kLayoutBinary64 (two fp32 boxes), kLayoutWide128 (four fp32 boxes), kLayoutWide64Q (four 8-bit boxes, default),
// kLayoutWide48Q (the same boxes, implied references).
// Closest hit: children are visited nearest first, the others are pushed so that they pop in ascending entry distance
// (order fixed by a 5-comparator network); any hit: farthest first (same network on the negated distance).  No re-cull on pop.  A leaf reference holds 1..8 consecutive triangles.
// Every step makes exactly ONE memory round trip: a lane first fetches its next item -- the node, or the next
// triangle(s) of its current leaf -- with one batch of 16-byte loads issued together, then branches into box or
// triangle tests (both branch-free).
__constant__ uint32_t g_pool_chunk = 256;   // rays per pool grab (RT3_OPT_POOL_CHUNK)
__constant__ uint32_t g_refill_lanes = 12;  // tuning knob (RT3_OPT_EXTEND_VARIANT)
void set_refill_lanes(uint32_t v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_refill_lanes), &v, 4); }
void set_pool_chunk(uint32_t v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pool_chunk), &v, 4); }
static unsigned g_trace_max_blocks = kExtendMaxBlocks;  // persistent traversal workgroups per launch (RT3_OPT_TRACE_BLOCKS)
void set_trace_blocks(uint32_t v) { g_trace_max_blocks = v; }

constexpr uint32_t kTopNodes = kTopCacheNodes;  // 8 KiB; ONE constant (rt3_internal.hpp) sizes the builder's array and the LDS copies
// copies the builder's top-of-tree array into LDS (kernel-uniform: either every thread of every block does, or none)
__device__ __forceinline__ bool load_top(float4* s_top, const float4* __restrict__ top, uint32_t n_top) {
    if (top == nullptr || n_top == 0u) return false;
    n_top = n_top < kTopNodes ? n_top : kTopNodes;  // never past the LDS array, whatever the host read back
    for (uint32_t i = threadIdx.x; i < 4u * n_top; i += kExtendBlock) s_top[i] = top[i];
    __syncthreads();
    return true;
}

// slab test of a quantised child whose plane bytes arrive near-first: p = {near.x, near.y, near.z, far.x}, q = {far.y, far.z, -, -}
// (v_perm_b32 by the ray's sign selectors).  The same maxima / minima as slab_test_q, minus the six that ordered each axis' pair.
__device__ __forceinline__ bool slab_test_sorted(uint32_t p, uint32_t q, V3 A, V3 B, float tmin, float tbest, float& tn_out) {
    const float nx = __builtin_fmaf((float)(p & 0xFFu), A.x, B.x), ny = __builtin_fmaf((float)((p >> 8) & 0xFFu), A.y, B.y),
                nz = __builtin_fmaf((float)((p >> 16) & 0xFFu), A.z, B.z), fx = __builtin_fmaf((float)(p >> 24), A.x, B.x),
                fy = __builtin_fmaf((float)(q & 0xFFu), A.y, B.y), fz = __builtin_fmaf((float)((q >> 8) & 0xFFu), A.z, B.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, tmin));
    const float tf = __builtin_fminf(__builtin_fminf(fx, fy), __builtin_fminf(fz, tbest));
    tn_out = tn;
    return tn <= tf;
}

struct LaneRay {  // traversal state of the ray a lane currently owns
    V3 o, d, inv;
    float tmin, inv_dd;  // inv_dd = 1 / d.d (the triangle test makes no unit-length assumption)
    Hit best;
    uint32_t cur, leaf_k, index, steps, cn, ct, cl;  // cn / ct / cl: node visits, triangle tests, node visits served by the LDS copy (COUNT)
    int sp;
    float pay0, pay1;  // shadow rays of the path tracer's own queue: two words of payload ride in the .w of the two ray records,
    float pay2, pay3;  // two more ({blue, path id}) in an 8-byte record fetched WITH the ray: at the end of the walk nothing is left to wait for
    uint32_t sel_p0, sel_q0, sel_p1, sel_q1;  // default layout: v_perm_b32 selectors that put a child's NEAR planes first (see the box block)
};

// MODE 0: closest hit over one queue; 1: any hit over one queue; 2: both queues in one walk -- the lanes of a wave take
// extension rays (closest hit) until that pool is dry and shadow rays (any hit) from then on, so the two kinds share a
// wave for a while and no lane waits for the wave's last extension ray before it starts on shadow rays.
template <int MODE, bool COUNT, int LAYOUT, typename Finish>
__device__ __forceinline__ void trace_stream(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             const float* __restrict__ rays_a, size_t stride, uint32_t n_a, uint32_t* __restrict__ work_counter_a,
                                             uint32_t* __restrict__ lds, Finish finish, bool any_payload = false,
                                             const float* __restrict__ rays_b = nullptr, uint32_t n_b = 0, uint32_t* __restrict__ work_counter_b = nullptr,
                                             bool ext_payload = false, const float4* top_lds = nullptr, bool use_top = false,
                                             const float2* __restrict__ any_contrib = nullptr) {
    // any_payload: the any-hit rays come from k_shade's shadow queue, where every ray has the range (kRayTMin, kBackgroundDepth):
    // the two .w slots of its record carry payload (two contribution channels) instead of tmin / tmax -- 16 bytes less per ray.
    // ext_payload: likewise for the extension rays of the path tracer's own queue (.w = the path's pdf and id, read by k_shade)
    const float* __restrict__ rays = rays_a;
    uint32_t n = n_a;
    uint32_t* __restrict__ work_counter = work_counter_a;
    bool second_pool = false;  // MODE 2: wave-uniform, true once this wave has moved on to the shadow queue
    bool lane_any = MODE == 1;
    constexpr bool WIDE = LAYOUT == kLayoutWide128;   // 8 x 16 B per fetch
    constexpr bool WIDEQ = LAYOUT == kLayoutWide64Q || LAYOUT == kLayoutWide48Q;  // quantised boxes
    constexpr bool C48 = LAYOUT == kLayoutWide48Q;  // 3 x 16 B per fetch: node and triangle records are both 48 B
    // ray pool: waves grab chunks of kPoolChunk consecutive rays from a per-launch counter (one returning atomic per
    // chunk: ~110 k per launch, ~20 / us, below the ~88 / us a single counter word sustains; 64-ray chunks were
    // atomic-bound, 1024 and more left a visible tail), so no wave idles at the end of a
    // launch while another still owns untouched rays
    uint32_t pool_next = 0, pool_end = 0;
    bool queue_empty = false;
    const uint32_t lane = __lane_id();
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    // a launch whose queue is mostly covered by the waves' static first chunks (a 1-spp frame: 2 M rays over 8192 waves) balances
    // better with chunks of half the size: 0.79 -> 0.72 ms per frame at 1080p, 1 spp, one bounce; long queues keep the full chunk
    const uint32_t kRefillLanes = g_refill_lanes,
                   kPoolChunk = (n_a < 2u * n_waves * g_pool_chunk && g_pool_chunk >= 128u) ? ((g_pool_chunk >> 1) & ~63u) : g_pool_chunk;
    // the first chunk of every wave is static (chunk number = global wave number): no atomic storm at launch, when all the
    // waves of the grid would hit the cursor at once (8192 returning atomics on one word ~ 0.1 ms); the cursor counts the
    // chunks handed out after those
    // (wave-uniform, but only the hardware knows: without the readfirstlane the pool cursors live in vector registers)
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    bool first_chunk = true;
    uint32_t spill[kSpill];
    LaneRay r;
    r.o = r.d = r.inv = v3(0.0f, 0.0f, 0.0f);
    r.tmin = 0.0f;
    r.inv_dd = 1.0f;
    r.best = Hit{0.0f, 0.0f, 0.0f, kMiss};
    r.cur = r.leaf_k = r.index = r.steps = r.cn = r.ct = r.cl = 0u;
    r.sp = 0;
    r.pay0 = r.pay1 = r.pay2 = r.pay3 = 0.0f;
    r.sel_p0 = r.sel_q0 = r.sel_p1 = r.sel_q1 = 0u;
    bool busy = false;
    for (;;) {
        // ---- refill idle lanes from the pool
        const unsigned long long m_idle = __ballot(!busy);
        if (pool_next >= pool_end && !queue_empty && m_idle != 0ull) {  // grab the next chunk (wave-uniform)
            uint32_t base = 0;
            if (first_chunk) {
                base = wave_id * kPoolChunk;
                first_chunk = false;
            } else {
                if (lane == 0) base = atomicAdd(work_counter, kPoolChunk);
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned long long b64 = (unsigned long long)n_waves * kPoolChunk + base;  // may exceed 2^32 only past the end
                base = b64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)b64;
            }
            queue_empty = base >= n;
            pool_next = base < n ? base : n;
            pool_end = (n - pool_next) > kPoolChunk ? pool_next + kPoolChunk : n;
            if (MODE == 2 && queue_empty && !second_pool) {  // extension queue dry: this wave's idle lanes go on with shadow rays
                second_pool = true;
                rays = rays_b;
                n = n_b;
                work_counter = work_counter_b;
                first_chunk = true;
                queue_empty = false;
                pool_next = pool_end = 0;
                continue;
            }
        }
        if (m_idle != 0ull && pool_next < pool_end && ((uint32_t)__popcll(m_idle) >= kRefillLanes || m_idle == ~0ull)) {
            const uint32_t idx = pool_next + (uint32_t)__popcll(m_idle & lanes_below);
            if (!busy && idx < pool_end) {
                const float4 ro = reinterpret_cast<const float4*>(rays)[idx], rd = reinterpret_cast<const float4*>(rays)[stride + idx];
                r.o = v3(ro.x, ro.y, ro.z);
                r.d = v3(rd.x, rd.y, rd.z);
                const bool payload = (any_payload && (MODE == 1 || (MODE == 2 && second_pool))) || (ext_payload && (MODE == 0 || (MODE == 2 && !second_pool)));
                r.tmin = payload ? kRayTMin : ro.w;
                r.best = Hit{payload ? kBackgroundDepth : rd.w, 0.0f, 0.0f, kMiss};
                r.pay0 = ro.w;
                r.pay1 = rd.w;
                if (any_contrib != nullptr && (MODE == 1 || (MODE == 2 && second_pool))) {
                    const float2 c2 = any_contrib[idx];
                    r.pay2 = c2.x;
                    r.pay3 = c2.y;
                }
                r.inv = v3(guarded_inverse(r.d.x), guarded_inverse(r.d.y), guarded_inverse(r.d.z));
                r.inv_dd = 1.0f / dot_fma(r.d, r.d);
                if (LAYOUT == kLayoutWide64Q) {
                    // A child's six plane bytes are {lo.xyz, hi.xyz} at bytes 0..5 (children 0, 2) or 2..7 (children 1, 3) of a word pair.
                    // Which of lo / hi is the NEAR plane of an axis is the sign of the ray's direction there, the same for every node:
                    // two v_perm_b32 per child pull {near.x, near.y, near.z, far.x} and {far.y, far.z} out, and the six min / max that
                    // ordered each pair are gone.  Same values: fma(q, A, B) is monotone in q, increasing for A >= 0, decreasing below,
                    // and no operand can be NaN (positions are bounded at upload, rays are checked finite, inverses are guarded).
                    const uint32_t sx = r.inv.x < 0.0f ? 1u : 0u, sy = r.inv.y < 0.0f ? 1u : 0u, sz = r.inv.z < 0.0f ? 1u : 0u;
                    const uint32_t nx = 3u * sx, ny = 1u + 3u * sy, nz = 2u + 3u * sz, fx = 3u - 3u * sx, fy = 4u - 3u * sy, fz = 5u - 3u * sz;
                    r.sel_p0 = nx | (ny << 8) | (nz << 16) | (fx << 24);
                    r.sel_q0 = fy | (fz << 8);
                    r.sel_p1 = r.sel_p0 + 0x02020202u;
                    r.sel_q1 = r.sel_q0 + 0x00000202u;
                }
                r.cur = (LAYOUT == kLayoutWide64Q && use_top) ? kTopFlag : 0u;  // the root: slot 0 of the LDS copy, or node 0
                r.leaf_k = 0u;
                r.sp = 0;
                r.index = idx;
                r.steps = 0u;
                r.cn = r.ct = r.cl = 0u;
                busy = true;
                if (MODE == 2) lane_any = second_pool;
                // a ray with a non-finite origin or direction (NaN camera, a zero-length shading normal upstream) misses: with NaNs every
                // slab test of the min/max form passes and the ray would walk the whole tree
                const float kMaxF = 3.4028234663852886e38f;
                const bool finite_ray = fabsf(r.o.x) <= kMaxF && fabsf(r.o.y) <= kMaxF && fabsf(r.o.z) <= kMaxF && fabsf(r.d.x) <= kMaxF &&
                                        fabsf(r.d.y) <= kMaxF && fabsf(r.d.z) <= kMaxF;
                if (nodes == nullptr || !finite_ray) {  // empty scene: everything misses
                    finish(r.index, r.best, 0u, 0u, 0u, MODE == 2 ? lane_any : MODE == 1, r.pay0, r.pay1, r.pay2, r.pay3);
                    busy = false;
                }
            }
            const uint32_t taken = (uint32_t)__popcll(m_idle);
            pool_next = pool_next + taken < pool_end ? pool_next + taken : pool_end;
        }
        if (__ballot(busy) == 0ull && pool_next >= pool_end && queue_empty) break;
        // ---- one traversal step.  (One region under `if (busy)` and a single way back to the loop header: with `continue`s in front of it
        // the compiler copied the eight registers of the walk's state aside at the top of every step and back at its end.)
        if (busy) {
        const bool ANY = MODE == 2 ? lane_any : MODE == 1;  // compile-time constant for MODE 0 / 1, per lane for MODE 2
        const bool is_leaf = (r.cur & 0x80000000u) != 0u;
        const uint32_t first = r.cur & 0x0FFFFFFFu, cnt = ((r.cur >> 28) & 7u) + 1u;
        const bool cached = LAYOUT == kLayoutWide64Q && !is_leaf && (r.cur & kTopFlag) != 0u;  // a top-of-tree node held in LDS
        const float4* p = is_leaf ? tris + 3 * (size_t)(first + r.leaf_k) : nodes + (WIDE ? 8 : (C48 ? kC48Stride : 4)) * (size_t)(cached ? 0u : r.cur);
        // one batch of loads (the triangle array carries 128 B of slack so that over-reading a leaf is in bounds)
        float4 q0, q1, q2, q3, q4, q5, q6, q7;
        if (LAYOUT == kLayoutWide64Q && cached) {
            const float4* t = top_lds + 4 * (r.cur & 0xFFFFu);
            q0 = t[0];
            q1 = t[1];
            q2 = t[2];
            q3 = t[3];
        } else {
            q0 = p[0];
            q1 = p[1];
            q2 = p[2];
            q3 = C48 ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : p[3];
        }
        if (WIDE) {
            q4 = p[4];
            q5 = p[5];
            q6 = p[6];
            q7 = p[7];
        }
        // pin the fetched registers: without this LLVM sinks the loads only one branch needs into that branch, which
        // turns one memory round trip per step into two
        pin(q0); pin(q1); pin(q2);
        if (!C48) pin(q3);
        if (WIDE) { pin(q4); pin(q5); pin(q6); pin(q7); }
        bool pop = false, done = false;
        const V3 o = r.o, d = r.d, inv = r.inv;
        const float tmin = r.tmin;
        if (is_leaf) {
            if (COUNT) r.ct++;
            tri_test_nb(q0, q1, q2, o, d, r.inv_dd, tmin, r.best);
            r.leaf_k++;
            if (WIDE) {  // the 128 B fetch holds a second triangle
                if (r.leaf_k < cnt && !(ANY && r.best.prim != kMiss)) {
                    if (COUNT) r.ct++;
                    tri_test_nb(q3, q4, q5, o, d, r.inv_dd, tmin, r.best);
                    r.leaf_k++;
                }
            }
            if (ANY && r.best.prim != kMiss) done = true;
            if (r.leaf_k >= cnt) {
                r.leaf_k = 0;
                pop = true;
            }
        } else {
            if (COUNT) {
                r.cn++;
                r.cl += cached ? 1u : 0u;
            }
            if (WIDE || WIDEQ) {
                const float kInf = __builtin_huge_valf();
                float t0, t1, t2, t3;
                uint32_t r0, r1, r2, r3;
                bool h0, h1, h2, h3;
                if (WIDEQ) {
                    // the dequantisation is folded into the ray: a child plane sits at org + q * 2^(e-127), so its ray parameter is
                    // q * (step * inv) + (org - o) * inv = fma(q, A, B): one v_cvt_f32_ubyte + one v_fma_f32 per plane.
                    // Bytes 6k..6k+5 of words 4..9 hold child k {lo.xyz, hi.xyz}.
                    const uint32_t ex = __float_as_uint(q0.w);
                    // (the 64-byte node holds its steps as floats -- word 3, words 14 / 15; the 48-byte one as three exponent bytes)
                    const V3 A = C48 ? v3(__uint_as_float((ex & 0xFFu) << 23) * inv.x, __uint_as_float(((ex >> 8) & 0xFFu) << 23) * inv.y,
                                          __uint_as_float(((ex >> 16) & 0xFFu) << 23) * inv.z)
                                     : v3(q0.w * inv.x, q3.z * inv.y, q3.w * inv.z);
                    const V3 B = v3((q0.x - o.x) * inv.x, (q0.y - o.y) * inv.y, (q0.z - o.z) * inv.z);
                    const uint32_t w0 = __float_as_uint(q1.x), w1 = __float_as_uint(q1.y), w2 = __float_as_uint(q1.z), w3 = __float_as_uint(q1.w),
                                   w4 = __float_as_uint(q2.x), w5 = __float_as_uint(q2.y);
                    if (C48) {
                        // implied references: internal children count up from node_base, leaf triangles from tri_base
                        const uint32_t wa = __float_as_uint(q2.z), wb = __float_as_uint(q2.w);
                        const uint32_t m0 = (ex >> 24) & 15u, m1 = ex >> 28, m2 = wa >> 28, m3 = wb >> 28;
                        uint32_t nb = wa & 0x0FFFFFFFu, tb = wb & 0x0FFFFFFFu;
#define RT3_REF48(m, out)                                                         \
    {                                                                             \
        const bool in_ = (m) == 0u, lf_ = ((m)&8u) != 0u;                         \
        out = in_ ? nb : (lf_ ? (0x80000000u | (((m)&7u) << 28) | tb) : kEmptySlot); \
        nb += in_ ? 1u : 0u;                                                      \
        tb += lf_ ? ((m)&7u) + 1u : 0u;                                           \
    }
                        RT3_REF48(m0, r0)
                        RT3_REF48(m1, r1)
                        RT3_REF48(m2, r2)
                        RT3_REF48(m3, r3)
#undef RT3_REF48
                    } else {
                        r0 = __float_as_uint(q2.z);
                        r1 = __float_as_uint(q2.w);
                        r2 = __float_as_uint(q3.x);
                        r3 = __float_as_uint(q3.y);
                    }
#define RT3_Q(w, b) ((float)(((w) >> (8 * (b))) & 0xFFu))
#define RT3_SLABQ(lx, ly, lz, hx, hy, hz, tn) \
    slab_test_q(__builtin_fmaf(lx, A.x, B.x), __builtin_fmaf(hx, A.x, B.x), __builtin_fmaf(ly, A.y, B.y), __builtin_fmaf(hy, A.y, B.y), \
                __builtin_fmaf(lz, A.z, B.z), __builtin_fmaf(hz, A.z, B.z), tmin, r.best.t, tn)
                    if (LAYOUT == kLayoutWide64Q) {
#define RT3_SLABS(wlo, whi, selp, selq, tn)                                                                                                    \
    slab_test_sorted(__builtin_amdgcn_perm(whi, wlo, selp), __builtin_amdgcn_perm(whi, wlo, selq), A, B, tmin, r.best.t, tn)
                        h0 = RT3_SLABS(w0, w1, r.sel_p0, r.sel_q0, t0) & (r0 != kEmptySlot);
                        h1 = RT3_SLABS(w1, w2, r.sel_p1, r.sel_q1, t1) & (r1 != kEmptySlot);
                        h2 = RT3_SLABS(w3, w4, r.sel_p0, r.sel_q0, t2) & (r2 != kEmptySlot);
                        h3 = RT3_SLABS(w4, w5, r.sel_p1, r.sel_q1, t3) & (r3 != kEmptySlot);
#undef RT3_SLABS
                    } else {
                    h0 = RT3_SLABQ(RT3_Q(w0, 0), RT3_Q(w0, 1), RT3_Q(w0, 2), RT3_Q(w0, 3), RT3_Q(w1, 0), RT3_Q(w1, 1), t0) & (r0 != kEmptySlot);
                    h1 = RT3_SLABQ(RT3_Q(w1, 2), RT3_Q(w1, 3), RT3_Q(w2, 0), RT3_Q(w2, 1), RT3_Q(w2, 2), RT3_Q(w2, 3), t1) & (r1 != kEmptySlot);
                    h2 = RT3_SLABQ(RT3_Q(w3, 0), RT3_Q(w3, 1), RT3_Q(w3, 2), RT3_Q(w3, 3), RT3_Q(w4, 0), RT3_Q(w4, 1), t2) & (r2 != kEmptySlot);
                    h3 = RT3_SLABQ(RT3_Q(w4, 2), RT3_Q(w4, 3), RT3_Q(w5, 0), RT3_Q(w5, 1), RT3_Q(w5, 2), RT3_Q(w5, 3), t3) & (r3 != kEmptySlot);
                    }
#undef RT3_Q
#undef RT3_SLABQ
                } else {
                    r0 = __float_as_uint(q1.z);
                    r1 = __float_as_uint(q3.z);
                    r2 = __float_as_uint(q5.z);
                    r3 = __float_as_uint(q7.z);
                    h0 = slab_test_hw(v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), o, inv, tmin, r.best.t, t0) & (r0 != kEmptySlot);
                    h1 = slab_test_hw(v3(q2.x, q2.y, q2.z), v3(q2.w, q3.x, q3.y), o, inv, tmin, r.best.t, t1) & (r1 != kEmptySlot);
                    h2 = slab_test_hw(v3(q4.x, q4.y, q4.z), v3(q4.w, q5.x, q5.y), o, inv, tmin, r.best.t, t2) & (r2 != kEmptySlot);
                    h3 = slab_test_hw(v3(q6.x, q6.y, q6.z), v3(q6.w, q7.x, q7.y), o, inv, tmin, r.best.t, t3) & (r3 != kEmptySlot);
                }
                // sort key: the entry distance for the closest hit (nearest child first); its NEGATIVE for any-hit rays
                // (farthest child first).  Occlusion does not depend on the order, but a shadow ray starts on a surface whose
                // neighbourhood it only grazes and is usually blocked far away (ceiling, opposite wall): far-first reaches that
                // occluder in ~35 % fewer node visits than slot order.  Non-entered slots carry +inf and sink to the end.
                Cand c0{h0 ? (ANY ? -t0 : t0) : kInf, h0 ? r0 : kEmptySlot}, c1{h1 ? (ANY ? -t1 : t1) : kInf, h1 ? r1 : kEmptySlot};
                Cand c2{h2 ? (ANY ? -t2 : t2) : kInf, h2 ? r2 : kEmptySlot}, c3{h3 ? (ANY ? -t3 : t3) : kInf, h3 ? r3 : kEmptySlot};
                const uint32_t nh = (uint32_t)h0 + (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3;
                // 5-comparator sorting network on the key
                cswap(c0, c1);
                cswap(c2, c3);
                cswap(c0, c2);
                cswap(c1, c3);
                cswap(c1, c2);
                if (nh > 3) {
                    if (r.sp < kLdsStack) lds[r.sp * kExtendBlock] = c3.ref;
                    else spill[r.sp - kLdsStack] = c3.ref;
                    ++r.sp;
                }
                if (nh > 2) {
                    if (r.sp < kLdsStack) lds[r.sp * kExtendBlock] = c2.ref;
                    else spill[r.sp - kLdsStack] = c2.ref;
                    ++r.sp;
                }
                if (nh > 1) {
                    if (r.sp < kLdsStack) lds[r.sp * kExtendBlock] = c1.ref;
                    else spill[r.sp - kLdsStack] = c1.ref;
                    ++r.sp;
                }
                r.cur = c0.ref;
                pop = nh == 0;
            } else {
                float tn0, tn1;
                uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
                bool h0 = slab_test_hw(v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), o, inv, tmin, r.best.t, tn0) & (r0 != kEmptySlot);
                bool h1 = slab_test_hw(v3(q1.z, q1.w, q2.x), v3(q2.y, q2.z, q2.w), o, inv, tmin, r.best.t, tn1) & (r1 != kEmptySlot);
                bool near1 = tn1 < tn0;
                if (h0 & h1) {
                    uint32_t far = near1 ? r0 : r1;
                    if (r.sp < kLdsStack) lds[r.sp * kExtendBlock] = far;
                    else spill[r.sp - kLdsStack] = far;
                    ++r.sp;
                }
                r.cur = (h0 & h1) ? (near1 ? r1 : r0) : (h0 ? r0 : r1);
                pop = !(h0 | h1);
            }
        }
        if (pop && !done) {
            if (r.sp == 0) {
                done = true;
            } else {
                --r.sp;
                // two explicit paths: a pointer select here would turn the pop into a flat_load
                if (r.sp < kLdsStack) {
                    r.cur = lds[r.sp * kExtendBlock];
                } else {
                    r.cur = spill[r.sp - kLdsStack];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // kMaxSteps bounds the walk so that a corrupt tree can never hang the GPU (a valid tree visits < 2 n nodes)
        if (++r.steps >= kMaxSteps) done = true;
        if (done) {
            finish(r.index, r.best, r.cn, r.ct, r.cl, MODE == 2 ? lane_any : MODE == 1, r.pay0, r.pay1, r.pay2, r.pay3);
            busy = false;
        }
        }  // if (busy)
    }
}

// closest-hit over a ray queue.  rays: two float4 streams of `stride` records, {o.xyz, tmin} then {d.xyz, tmax};
// hits: one float4 {t, u, v, prim} per ray.  16-byte records are the widest coalesced access (1 KiB per wave instruction).
template <bool COUNT, int LAYOUT>
__global__ __launch_bounds__(kExtendBlock) void k_extend(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float4* __restrict__ top, uint32_t n_top,
                                                         const float* __restrict__ rays, size_t stride,
                                                         const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
                                                         float* __restrict__ hits, uint32_t* __restrict__ cnt_nodes,
                                                         uint32_t* __restrict__ cnt_tris, unsigned long long* __restrict__ totals,
                                                         uint32_t* __restrict__ work_counter, int payload, unsigned long long* __restrict__ lds_total) {
    __shared__ uint32_t stack[kLdsStack * kExtendBlock];
    __shared__ float4 s_top[4 * kTopNodes];
    const bool use_top = load_top(s_top, top, n_top);  // (the LDS array itself is passed on, never a selected pointer: a select would turn its reads into flat loads)
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    unsigned long long tot_n = 0, tot_t = 0, tot_l = 0;
    auto finish = [&](uint32_t i, const Hit& h, uint32_t cn, uint32_t ct, uint32_t cl, bool, float, float, float, float) {
        // one 16-byte record per ray: with persistent waves rays finish out of order, four SoA streams would be four
        // scattered partial-line writes
        reinterpret_cast<float4*>(hits)[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
        if (COUNT) {
            if (cnt_nodes) cnt_nodes[i] = cn;
            if (cnt_tris) cnt_tris[i] = ct;
            tot_n += cn;
            tot_t += ct;
            tot_l += cl;
        }
    };
    trace_stream<0, COUNT, LAYOUT>(nodes, tris, rays, stride, n, work_counter, stack + threadIdx.x, finish, false, nullptr, 0, nullptr, payload != 0, s_top, use_top);
    if (COUNT && totals) {
        atomicAdd(&totals[0], tot_n);
        atomicAdd(&totals[1], tot_t);
        if (lds_total) atomicAdd(lds_total, tot_l);
    }
}

// any-hit over the shadow queue; unoccluded rays add their contribution to the path's radiance slot.
// If `occluded_out` != nullptr the kernel only reports occlusion (rt3_trace_rays).
template <bool COUNT, int LAYOUT>
__global__ __launch_bounds__(kExtendBlock) void k_shadow(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float4* __restrict__ top, uint32_t n_top,
                                                         const float* __restrict__ rays, size_t stride,
                                                         const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
                                                         const float* __restrict__ contrib, const uint32_t* __restrict__ pid,
                                                         float* __restrict__ lacc, size_t lstride,
                                                         uint32_t* __restrict__ occluded_out, uint32_t* __restrict__ cnt_nodes,
                                                         uint32_t* __restrict__ cnt_tris, unsigned long long* __restrict__ totals,
                                                         uint32_t* __restrict__ work_counter, unsigned long long* __restrict__ lds_total) {
    __shared__ uint32_t stack[kLdsStack * kExtendBlock];
    __shared__ float4 s_top[4 * kTopNodes];
    const bool use_top = load_top(s_top, top, n_top);  // (the LDS array itself is passed on, never a selected pointer: a select would turn its reads into flat loads)
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    unsigned long long tot_n = 0, tot_t = 0, tot_l = 0;
    trace_stream<1, COUNT, LAYOUT>(
        nodes, tris, rays, stride, n, work_counter, stack + threadIdx.x,
        [&](uint32_t i, const Hit& h, uint32_t cn, uint32_t ct, uint32_t cl, bool, float c_r, float c_g, float c_b, float c_pid) {
            if (occluded_out) {
                occluded_out[i] = h.prim != kMiss ? 1u : 0u;
            } else if (h.prim == kMiss) {
                // red and green rode with the ray, {blue, path id} arrived with it: one 16-byte read-modify-write per path, and the
                // wave waits for ONE round trip here (it used to be two: the id first, the slot after)
                float4* L = reinterpret_cast<float4*>(lacc) + __float_as_uint(c_pid);
                float4 v = *L;
                *L = make_float4(v.x + c_r, v.y + c_g, v.z + c_b, 0.0f);
            }
            if (COUNT) {
                if (cnt_nodes) cnt_nodes[i] = cn;
                if (cnt_tris) cnt_tris[i] = ct;
                tot_n += cn;
                tot_t += ct;
                tot_l += cl;
            }
        },
        occluded_out == nullptr, nullptr, 0, nullptr, false, s_top, use_top, occluded_out == nullptr ? reinterpret_cast<const float2*>(contrib) : nullptr);
    if (COUNT && totals) {
        atomicAdd(&totals[0], tot_n);
        atomicAdd(&totals[1], tot_t);
        if (lds_total) atomicAdd(lds_total, tot_l);
    }
}

// One launch per bounce for BOTH ray kinds: the lanes of every wave take extension rays (closest hit) until that queue is dry and
// shadow rays (any hit) from then on -- no grid-wide barrier and no per-wave drain in between -- a launch boundary idles the machine while the last
// waves finish (each k_extend / k_shadow pair cost one such drain more), which matters most when the frame is split
// over several GPUs and every launch is 1/N as long.  totals (counting mode): {rays, nodes, tris} x {closest, any}.
template <bool COUNT, int LAYOUT>
__global__ __launch_bounds__(kExtendBlock) void k_trace(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float4* __restrict__ top, uint32_t n_top,
                                                        const float* __restrict__ ext_rays, const float* __restrict__ sh_rays, size_t stride,
                                                        const uint32_t* __restrict__ ext_count, const uint32_t* __restrict__ sh_count,
                                                        float* __restrict__ hits, const float* __restrict__ contrib, float* __restrict__ lacc,
                                                        unsigned long long* __restrict__ totals, uint32_t* __restrict__ work_ext,
                                                        uint32_t* __restrict__ work_sh) {
    __shared__ uint32_t stack[kLdsStack * kExtendBlock];
    __shared__ float4 s_top[4 * kTopNodes];
    const bool use_top = load_top(s_top, top, n_top);  // (the LDS array itself is passed on, never a selected pointer: a select would turn its reads into flat loads)
    const uint32_t n_ext = *ext_count, n_sh = *sh_count;
    unsigned long long en = 0, et = 0, sn = 0, stt = 0, el = 0, sl = 0;
    trace_stream<2, COUNT, LAYOUT>(
        nodes, tris, ext_rays, stride, n_ext, work_ext, stack + threadIdx.x,
        [&](uint32_t i, const Hit& h, uint32_t cn, uint32_t ct, uint32_t cl, bool any, float c_r, float c_g, float c_b, float c_pid) {
            if (!any) {
                reinterpret_cast<float4*>(hits)[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
            } else if (h.prim == kMiss) {
                float4* L = reinterpret_cast<float4*>(lacc) + __float_as_uint(c_pid);  // {blue, path id} came with the ray
                float4 v = *L;
                *L = make_float4(v.x + c_r, v.y + c_g, v.z + c_b, 0.0f);
            }
            if (COUNT) {
                en += any ? 0u : cn;
                et += any ? 0u : ct;
                sn += any ? cn : 0u;
                stt += any ? ct : 0u;
                el += any ? 0u : cl;
                sl += any ? cl : 0u;
            }
        },
        true, sh_rays, n_sh, work_sh, true, s_top, use_top, reinterpret_cast<const float2*>(contrib));
    if (COUNT && totals) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&totals[0], (unsigned long long)n_ext);
            atomicAdd(&totals[3], (unsigned long long)n_sh);
        }
        atomicAdd(&totals[1], en);
        atomicAdd(&totals[2], et);
        atomicAdd(&totals[4], sn);
        atomicAdd(&totals[5], stt);
        atomicAdd(&totals[6], el);  // (= d_totals[10], [11]: node visits served by the LDS copy, closest / any)
        atomicAdd(&totals[7], sl);
    }
}

// ------------------------------------------------------------------------------------------------ gbuffer pass
// gbuffer.slang:8-12 : primary rays for the pixels this rank owns (pixel list is in tile / Z-curve order)
__global__ void k_raygen(GConstDev g, const uint32_t* __restrict__ pixels, uint32_t npix, float* __restrict__ rays, size_t stride) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i];
        V3 o, d;
        primary_ray(g, xy & 0xFFFFu, xy >> 16, o, d);
        reinterpret_cast<float4*>(rays)[i] = make_float4(o.x, o.y, o.z, 0.0f);                         // TMin, gbuffer_helpers.slang:100
        reinterpret_cast<float4*>(rays)[stride + i] = make_float4(d.x, d.y, d.z, kBackgroundDepth);    // TMax, :101
    }
}
// gbuffer.slang:15-20
__global__ void k_gbuffer(SceneDev sc, const uint32_t* __restrict__ pixels, uint32_t npix, uint32_t width,
                          const float* __restrict__ hits, size_t stride, uint4* __restrict__ gbuffer, float* __restrict__ depth) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i];
        size_t pi = (size_t)(xy >> 16) * width + (xy & 0xFFFFu);
        const float4 hrec = reinterpret_cast<const float4*>(hits)[i];
        uint32_t prim = __float_as_uint(hrec.w);
        if (prim == kMiss) {
            depth[pi] = kBackgroundDepth;
        } else {
            Surface s = hit_info(sc, prim, hrec.y, hrec.z);
            gbuffer[pi] = gbuffer_pack(s);
            depth[pi] = hrec.x;
        }
    }
}

// ------------------------------------------------------------------------------------------------ shading
// Queue append for a whole workgroup: every wave ballots its lanes, the wave totals meet in LDS and ONE returning
// atomic per workgroup and queue reserves the range (a single counter word sustains only ~88 returning atomics per
// microsecond chip-wide, so one atomic per wave made k_shade atomic-bound: 1 M atomics per launch ~ 10 ms).
// Order is preserved inside the workgroup.  Must be reached by all threads of the block.
constexpr int kShadeBlock = 512;
struct BlockAppend {
    uint32_t ext, sh;
};
__device__ __forceinline__ BlockAppend block_append2(bool want_ext, bool want_sh, unsigned long long* pair /* {ext count, shadow count} */,
                                                     uint32_t* lds /* 2 * (waves + 1) words */) {
    constexpr int kWaves = kShadeBlock / 64;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long m_ext = __ballot(want_ext), m_sh = __ballot(want_sh);
    if (lane == 0) {
        lds[wave] = (uint32_t)__popcll(m_ext);
        lds[kWaves + 1 + wave] = (uint32_t)__popcll(m_sh);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t te = 0, ts = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            uint32_t ce = lds[w], cs = lds[kWaves + 1 + w];
            lds[w] = te;
            lds[kWaves + 1 + w] = ts;
            te += ce;
            ts += cs;
        }
        // both queue sizes live in one 8-byte word and move with ONE returning atomic (they used to be two atomics on the same
        // cache line, i.e. on the same L2 atomic unit: the unit's throughput is what a 256-thread block size ran into)
        unsigned long long old = 0ull;
        if (te | ts) old = atomicAdd(pair, (unsigned long long)te | ((unsigned long long)ts << 32));
        lds[kWaves] = (uint32_t)old;
        lds[2 * kWaves + 1] = (uint32_t)(old >> 32);
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    BlockAppend r;
    r.ext = lds[kWaves] + lds[wave] + (uint32_t)__popcll(m_ext & below);
    r.sh = lds[2 * kWaves + 1] + lds[kWaves + 1 + wave] + (uint32_t)__popcll(m_sh & below);
    return r;  // no third barrier: the caller alternates between two lds buffers from one loop iteration to the next
}

struct ShadeArgs {
    GConstDev g;
    SceneDev sc;
    const uint32_t* pixels;  // x | y << 16, this rank's pixels in render order
    const uint2* pixbn;      // the same list with each pixel's blue-noise word beside it
    uint32_t npix, width;
    FastDiv npix_div;        // path id = sample_in_batch * npix + pixel_index
    uint32_t s0;             // first sample index of this batch
    uint32_t bounce;         // b
    // FIRST: gbuffer images
    const uint4* gbuffer;
    const float* depth;
    // !FIRST: input queue
    const float* in_rays;
    const float* in_hits;
    const float* in_T;       // throughput: three planes of `stride` floats (the path's pdf and id ride in the .w of its two ray records)
    const uint32_t* in_count;
    uint32_t n_first;        // FIRST: npix * samples_in_batch
    // outputs
    float* out_rays;
    float* out_T;
    uint32_t* out_count;
    float* sh_rays;
    float* sh_contrib;
    uint32_t* sh_count;
    float* lacc;             // float4 {r, g, b, -} per path id
    size_t stride;
};

// refrence_mode.slang:28-57 for one bounce of every live path
// GLDS: the flattened-geometry table (80-byte entries, at most kShadeGeomsLds of them) is staged in LDS, so that a hit's material and
// normal matrix cost an LDS read behind the shading record instead of a second dependent global gather
template <bool FIRST, bool GLDS>
// 6 waves per SIMD (80 VGPRs, 32 bytes of scratch): the kernel lives off memory-level parallelism -- 28.8 -> 27.7 ms against the
// compiler's own choice of 93 VGPRs (4 waves with 512-thread blocks)
__global__ __launch_bounds__(kShadeBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_shade(ShadeArgs a) {
    __shared__ uint32_t append_lds[2][2 * (kShadeBlock / 64 + 1)];  // double-buffered: see block_append2
    __shared__ ShadeGeomDev s_geoms[GLDS ? kShadeGeomsLds : 1];
    if (GLDS) {
        const uint32_t words = (a.sc.n_geoms < kShadeGeomsLds ? a.sc.n_geoms : kShadeGeomsLds) * (uint32_t)(sizeof(ShadeGeomDev) / 4);
        for (uint32_t k = threadIdx.x; k < words; k += kShadeBlock) reinterpret_cast<uint32_t*>(s_geoms)[k] = reinterpret_cast<const uint32_t*>(a.sc.shade_geoms)[k];
        __syncthreads();
    }
    uint32_t parity = 0;
    // the marginal sky tables (one guide word + one CDF value per row) are the first two links of every light sample's chain of
    // dependent lookups: staged in LDS they cost ~100 cycles each instead of an L1/L2 round trip
    constexpr uint32_t kMargRows = 2048;
    __shared__ uint32_t s_guide_marg[kMargRows];
    __shared__ float s_cdf_marg[kMargRows + 4];
    const bool marg_in_lds = (a.g.pad[0] & RT3_FLAG_NEE_SKY) && a.sc.sky != nullptr && a.sc.sky_h <= kMargRows;
    if (marg_in_lds) {
        for (uint32_t k = threadIdx.x; k < a.sc.sky_h; k += kShadeBlock) s_guide_marg[k] = a.sc.guide_marg[k];
        for (uint32_t k = threadIdx.x; k < a.sc.sky_h + 4u; k += kShadeBlock) s_cdf_marg[k] = a.sc.cdf_marg[k];
        __syncthreads();
    }
    const float* cdf_marg = marg_in_lds ? s_cdf_marg : a.sc.cdf_marg;
    const uint32_t* guide_marg = marg_in_lds ? s_guide_marg : a.sc.guide_marg;
    const GConstDev& g = a.g;
    const uint32_t flags = g.pad[0], B = g.bounces, b = a.bounce;
    const uint32_t dims = flags ? 8u : 2u;
    const bool nee = (flags & RT3_FLAG_NEE_SKY) && a.sc.sky != nullptr;
    const bool bnz = (flags & RT3_FLAG_BLUENOISE) && a.sc.bluenoise != nullptr;
    const bool spec = (flags & RT3_FLAG_SPECULAR) != 0u;
    const size_t S = a.stride;
    const uint32_t n = FIRST ? a.n_first : *a.in_count;
    const uint32_t n_round = (n + (kShadeBlock - 1)) & ~(uint32_t)(kShadeBlock - 1);  // whole workgroups iterate: barriers inside
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        bool active = i < n;
        uint32_t pid = 0;
        V3 o = v3(0, 0, 0), d = v3(0, 0, 1), T = v3(1, 1, 1);
        float t = 0.0f, pdf_b = 0.0f;
        Surface surf;
        surf.albedo = surf.emissive = v3(0, 0, 0);
        surf.normal = v3(0, 0, 1);
        surf.roughness = 1.0f;
        surf.metalness = 0.0f;
        uint32_t px = 0, py = 0, sample_in_batch = 0, bn = 0;
        HitRecord hrecord;
        hrecord.rec = make_uint4(0u, 0u, 0u, 0u);
        hrecord.prim = 0u;
        float hbu = 0.0f, hbv = 0.0f;
        float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (!FIRST && active) {  // ray records {o, pdf} + {d, path id}
            ro = reinterpret_cast<const float4*>(a.in_rays)[i];
            rd = reinterpret_cast<const float4*>(a.in_rays)[S + i];
        }
        if (active) {
            pid = FIRST ? i : __float_as_uint(rd.w);
            sample_in_batch = fast_div(a.npix_div, pid);
            const uint2 pb = a.pixbn[pid - sample_in_batch * a.npix];  // pixel and its blue-noise word in one load
            px = pb.x & 0xFFFFu;
            py = pb.x >> 16;
            bn = pb.y;
        }
        if (FIRST) {
            if (active) {
                size_t pi = (size_t)py * a.width + px;
                float d0 = a.depth[pi];
                if (d0 == kBackgroundDepth) {  // :18-21
                    reinterpret_cast<float4*>(a.lacc)[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    active = false;
                } else {
                    surf = gbuffer_unpack(a.gbuffer[pi]);  // :23
                    primary_ray(g, px, py, o, d);          // :30
                    t = d0;                                // :33
                }
            }
        } else if (active) {
            o = v3(ro.x, ro.y, ro.z);
            d = v3(rd.x, rd.y, rd.z);
            T = v3(a.in_T[i], a.in_T[S + i], a.in_T[2 * S + i]);
            pdf_b = ro.w;
            const float4 hrec = reinterpret_cast<const float4*>(a.in_hits)[i];
            uint32_t prim = __float_as_uint(hrec.w);
            if (prim == kMiss) {  // :37-40 ; sky through MIS (north_star)
                if (nee && pdf_b > 0.0f) {
                    float su, sv;
                    direction_to_equirect_uv(d, su, sv);
                    float pl;
                    V3 rad = sky_eval_and_pdf(a.sc, su, sv, pl);
                    float w = pdf_b / (pdf_b + pl);
                    float4* Lp = reinterpret_cast<float4*>(a.lacc) + pid;
                    float4 lv = *Lp;
                    *Lp = make_float4(lv.x + T.x * (rad.x * w), lv.y + T.y * (rad.y * w), lv.z + T.z * (rad.z * w), 0.0f);
                }
                active = false;
            } else {
                t = hrec.x;
                hbu = hrec.y;
                hbv = hrec.z;
                hrecord = hit_fetch(a.sc, prim);  // :55, first half: the loads are issued here, used after the light-sample gathers
            }
        }
        bool emit_shadow = false, emit_ext = false;
        V3 wl = v3(0, 1, 0), contrib = v3(0, 0, 0), nd = v3(0, 0, 1), Tn = T;
        float pdf_n = 0.0f;
        if (active) {
            uint32_t seed = rng_seed(px, py, g.frame);  // :25
            uint32_t sm = a.s0 + sample_in_batch;
            uint32_t base = (sm * B + b) * dims;
            float u0 = uniform_float(seed, base), u1 = uniform_float(seed, base + 1);  // :43
            if (bnz) {  // bn = bluenoise[(py % bn_h), (px % bn_w)], gathered once per pixel list (k_pixbn)
                u0 = bluenoise_shift(u0, bn & 0xFFu);
                u1 = bluenoise_shift(u1, (bn >> 8) & 0xFFu);
            }
            // the light sample first: its chain of dependent table gathers (guide -> CDF -> guide -> CDF -> texels) is then in
            // flight while the shading record arrives and the BSDF is set up and sampled below.  Samples below the horizon
            // of the surface (cos <= 0) are dropped before the radiance texels are fetched: those two cache lines are the
            // expensive part of a light sample.
            V3 rad = v3(0.0f, 0.0f, 0.0f);
            float pl = 0.0f, cosl = 0.0f;
            SkyPick pick;
            pick.u = pick.v = pick.sin_theta = 0.0f;
            pick.x = pick.y = 0;
            if (nee) {
                float ul0 = uniform_float(seed, base + 3), ul1 = uniform_float(seed, base + 4);
                if (bnz) {
                    ul0 = bluenoise_shift(ul0, (bn >> 16) & 0xFFu);
                    ul1 = bluenoise_shift(ul1, (bn >> 24) & 0xFFu);
                }
                pick = sky_sample_direction(a.sc, cdf_marg, guide_marg, ul0, ul1, wl);
            }
            if (!FIRST) {  // :55, second half (two calls, not a selected pointer: a select would turn the LDS reads into flat loads)
                if (GLDS) surf = hit_finish(a.sc, s_geoms, hrecord, hbu, hbv);
                else surf = hit_finish(a.sc, a.sc.shade_geoms, hrecord, hbu, hbv);
            }
            V3 N = surf.normal;
            if ((flags & RT3_FLAG_FACEFORWARD) && dot(N, d) > 0.0f) N = neg(N);
            if (nee) {
                cosl = dot(N, wl);
                if (cosl > 0.0f) sky_sample_radiance(a.sc, pick, rad, pl);
            }
            V3 b1, b2;
            build_orthonormal_basis(N, b1, b2);  // :44
            V3 wi = v3(0.0f, 0.0f, 1.0f), vop = surf.albedo, wo = v3(0.0f, 0.0f, 1.0f);
            float pdf_s = 0.0f;
            bool valid = true;
            Bsdf bs;
            const bool last = b == B - 1;  // the last vertex of a path emits no extension ray (:53): its BSDF sample is never used
            if (spec) {  // layered diffuse + GGX (brdf.slang:141-311)
                bs = bsdf_setup(surf.albedo, surf.roughness, surf.metalness);
                wo = v3(-(d.x * b1.x + d.y * b1.y + d.z * b1.z), -(d.x * b2.x + d.y * b2.y + d.z * b2.z), -(d.x * N.x + d.y * N.y + d.z * N.z));
                if (!last) {
                    float u2 = uniform_float(seed, base + 2);
                    valid = bsdf_sample(bs, wo, u0, u1, u2, wi, vop, pdf_s);
                }
            } else if (!last) {
                wi = diffuse_sample(u0, u1);  // :45
                pdf_s = wi.z * kInvPi;
            }
            o = v3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);  // :47
            // :50 radiance += ray_color * emissive.  x + (+0) == x exactly, so non-emitters skip the read-modify-write.
            if (FIRST) {  // L starts at 0 and T = 1: 0 + 1 * e == e
                reinterpret_cast<float4*>(a.lacc)[pid] = make_float4(T.x * surf.emissive.x, T.y * surf.emissive.y, T.z * surf.emissive.z, 0.0f);
            } else if (surf.emissive.x != 0.0f || surf.emissive.y != 0.0f || surf.emissive.z != 0.0f) {
                float4* Lp = reinterpret_cast<float4*>(a.lacc) + pid;
                float4 lv = *Lp;
                *Lp = make_float4(lv.x + T.x * surf.emissive.x, lv.y + T.y * surf.emissive.y, lv.z + T.z * surf.emissive.z, 0.0f);
            }
            if (nee) {
                if (cosl > 0.0f && pl > 0.0f) {
                    V3 fv;
                    float scale;
                    if (spec) {  // evaluate the layered BSDF towards the light; f cos / (p_light + p_bsdf)
                        float pproj;
                        bsdf_eval(bs, wo, v3(dot(wl, b1), dot(wl, b2), cosl), fv, pproj);
                        float pb = pproj * cosl;
                        scale = (b == B - 1) ? cosl / pl : cosl / (pl + pb);
                    } else {  // diffuse: f = albedo / pi folded into the scale
                        float pb = cosl * kInvPi;
                        fv = surf.albedo;
                        scale = (b == B - 1) ? (cosl * kInvPi) / pl : (cosl * kInvPi) / (pl + pb);
                    }
                    contrib = v3((T.x * fv.x) * (rad.x * scale), (T.y * fv.y) * (rad.y * scale), (T.z * fv.z) * (rad.z * scale));
                    emit_shadow = true;
                }
            }
            if (valid && !last) {  // an invalid specular sample (brdf.slang:227-229) ends the path
                nd = basis_apply(b1, b2, N, wi);  // :48
                pdf_n = pdf_s;
                Tn = T * vop;                     // :51 value_over_pdf (= albedo for the diffuse BRDF)
                emit_ext = true;                  // :53
            }
        }
        const BlockAppend slot = block_append2(emit_ext, emit_shadow, reinterpret_cast<unsigned long long*>(a.out_count), append_lds[parity]);  // out_count, sh_count: one pair
        parity ^= 1u;
        if (emit_shadow) {
            const uint32_t j = slot.sh;
            // 40 bytes per shadow ray: its range is always (kRayTMin, kBackgroundDepth), so the .w of the two ray records carry the
            // red and green contribution; blue and the path id follow in an 8-byte record
            reinterpret_cast<float4*>(a.sh_rays)[j] = make_float4(o.x, o.y, o.z, contrib.x);
            reinterpret_cast<float4*>(a.sh_rays)[S + j] = make_float4(wl.x, wl.y, wl.z, contrib.y);
            reinterpret_cast<float2*>(a.sh_contrib)[j] = make_float2(contrib.z, __uint_as_float(pid));
        }
        if (emit_ext) {
            const uint32_t j = slot.ext;
            // 44 bytes per extension ray: its range is always (kRayTMin, kBackgroundDepth) (:31), so the .w of its two ray records carry
            // the path's pdf and id; the throughput follows as three 4-byte planes (queue slots are contiguous per workgroup: coalesced)
            reinterpret_cast<float4*>(a.out_rays)[j] = make_float4(o.x, o.y, o.z, pdf_n);
            reinterpret_cast<float4*>(a.out_rays)[S + j] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pid));
            a.out_T[j] = Tn.x;
            a.out_T[S + j] = Tn.y;
            a.out_T[2 * S + j] = Tn.z;
        }
    }
}

// refrence_mode.slang:59-65 : radiance = (sum over samples, in sample order) / S, then blend with PrevLight
__global__ void k_accumulate(GConstDev g, const uint32_t* __restrict__ pixels, uint32_t npix, uint32_t width,
                             const float* __restrict__ depth, const float* __restrict__ lacc, size_t stride, uint32_t sb,
                             int first_batch, int last_batch, float* __restrict__ radsum, float4* __restrict__ light,
                             const float4* __restrict__ prev) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i];
        size_t pi = (size_t)(xy >> 16) * width + (xy & 0xFFFFu);
        if (depth[pi] == kBackgroundDepth) continue;
        float r = first_batch ? 0.0f : radsum[i], gg = first_batch ? 0.0f : radsum[npix + i], bb = first_batch ? 0.0f : radsum[2 * (size_t)npix + i];
        for (uint32_t s = 0; s < sb; s++) {
            size_t p = (size_t)s * npix + i;
            const float4 lv = reinterpret_cast<const float4*>(lacc)[p];
            r += lv.x;
            gg += lv.y;
            bb += lv.z;
        }
        if (!last_batch) {
            radsum[i] = r;
            radsum[npix + i] = gg;
            radsum[2 * (size_t)npix + i] = bb;
        } else {
            float fs = (float)g.samples;
            r = r / fs;
            gg = gg / fs;
            bb = bb / fs;
            if (g.blendfactor >= 1.0f) {
                light[pi] = make_float4(r, gg, bb, 0.0f);
            } else {
                float4 pv = prev[pi];
                float bf = g.blendfactor;
                light[pi] = make_float4(pv.x + (r - pv.x) * bf, pv.y + (gg - pv.y) * bf, pv.z + (bb - pv.z) * bf, 0.0f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ postprocess
// postprocess.slang:13-25
__device__ __forceinline__ float agx_contrast(float x) {
    float x2 = x * x, x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}
// postprocess.slang:27-88 (AGX_LOOK 2; pow base clamped at 0)
__device__ __forceinline__ V3 agx_tonemap(V3 c) {
    const float m[9] = {0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f, 0.0784335999999992f, 0.878468636469772f,
                        0.0784336f, 0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f};
    const float mi[9] = {1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f, -0.0980208811401368f, 1.15190312990417f,
                         -0.0980434501171241f, -0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f};
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    float v[3], w[3];
#pragma unroll
    for (int j = 0; j < 3; j++) v[j] = c.x * m[j] + c.y * m[3 + j] + c.z * m[6 + j];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float l = v[j] > 0.0f ? log2f(v[j]) : min_ev;
        l = fmin_sel(fmax_sel(l, min_ev), max_ev);
        l = (l - min_ev) / (max_ev - min_ev);
        v[j] = agx_contrast(l);
    }
    float luma = v[0] * 0.2126f + v[1] * 0.7152f + v[2] * 0.0722f;
#pragma unroll
    for (int j = 0; j < 3; j++) w[j] = luma + 1.1f * (powf(fmax_sel(v[j], 0.0f), 1.1f) - luma);
    return v3(w[0] * mi[0] + w[1] * mi[3] + w[2] * mi[6], w[0] * mi[1] + w[1] * mi[4] + w[2] * mi[7], w[0] * mi[2] + w[1] * mi[5] + w[2] * mi[8]);
}
// postprocess.slang:90-112
__global__ void k_postprocess(GConstDev g, SceneDev sc, const uint32_t* __restrict__ pixels, uint32_t npix, uint32_t width,
                              const float* __restrict__ depth, const float4* __restrict__ in, float4* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i], px = xy & 0xFFFFu, py = xy >> 16;
        size_t pi = (size_t)py * width + px;
        V3 col;
        if (depth[pi] != kBackgroundDepth) {
            float4 c = in[pi];
            col = v3(c.x, c.y, c.z);
        } else {
            V3 o, d;
            primary_ray(g, px, py, o, d);
            float su, sv;
            direction_to_equirect_uv(d, su, sv);
            col = sky_eval(sc, su, sv);
        }
        V3 r = agx_tonemap(col);
        out[pi] = make_float4(r.x, r.y, r.z, 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------ tiles
__global__ void k_pack_tiles(const uint32_t* __restrict__ pixels, uint32_t npix, uint32_t width, const float4* __restrict__ img,
                             float4* __restrict__ dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i];
        dst[i] = img[(size_t)(xy >> 16) * width + (xy & 0xFFFFu)];
    }
}
__global__ void k_unpack_tiles(const uint32_t* __restrict__ pixels, uint32_t npix, uint32_t width, const float4* __restrict__ src,
                               float4* __restrict__ img) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        uint32_t xy = pixels[i];
        img[(size_t)(xy >> 16) * width + (xy & 0xFFFFu)] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------ self-test
// Evaluates one device function per element so that tests can pin the device arithmetic against known answers
// (rt3_selftest_eval).  in / out are dense arrays of `in_w` / `out_w` 32-bit words per element.
__global__ void k_selftest(int op, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto F = [](uint32_t u) { return __uint_as_float(u); };
    auto U = [](float f) { return __float_as_uint(f); };
    switch (op) {
        case 0: out[i] = jenkins_hash(in[i]); break;
        case 1: out[i] = zcurve(in[2 * i], in[2 * i + 1]); break;
        case 2: out[i] = murmur3(in[2 * i], in[2 * i + 1]); break;
        case 3: out[i] = U(uniform_float(in[2 * i], in[2 * i + 1])); break;
        case 4: {
            const uint32_t* p = in + 11 * i;
            Surface s;
            s.albedo = v3(F(p[0]), F(p[1]), F(p[2]));
            s.emissive = v3(F(p[3]), F(p[4]), F(p[5]));
            s.normal = v3(F(p[6]), F(p[7]), F(p[8]));
            s.roughness = F(p[9]);
            s.metalness = F(p[10]);
            uint4 q = gbuffer_pack(s);
            out[4 * i] = q.x; out[4 * i + 1] = q.y; out[4 * i + 2] = q.z; out[4 * i + 3] = q.w;
            break;
        }
        case 5: {
            Surface s = gbuffer_unpack(make_uint4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]));
            uint32_t* o = out + 11 * i;
            o[0] = U(s.albedo.x); o[1] = U(s.albedo.y); o[2] = U(s.albedo.z);
            o[3] = U(s.emissive.x); o[4] = U(s.emissive.y); o[5] = U(s.emissive.z);
            o[6] = U(s.normal.x); o[7] = U(s.normal.y); o[8] = U(s.normal.z);
            o[9] = U(s.roughness); o[10] = U(s.metalness);
            break;
        }
        case 6: {
            V3 w = diffuse_sample(F(in[2 * i]), F(in[2 * i + 1]));
            out[3 * i] = U(w.x); out[3 * i + 1] = U(w.y); out[3 * i + 2] = U(w.z);
            break;
        }
        case 7: {
            V3 b1, b2;
            build_orthonormal_basis(v3(F(in[3 * i]), F(in[3 * i + 1]), F(in[3 * i + 2])), b1, b2);
            uint32_t* o = out + 6 * i;
            o[0] = U(b1.x); o[1] = U(b1.y); o[2] = U(b1.z); o[3] = U(b2.x); o[4] = U(b2.y); o[5] = U(b2.z);
            break;
        }
        case 8: {
            V3 r = agx_tonemap(v3(F(in[3 * i]), F(in[3 * i + 1]), F(in[3 * i + 2])));
            out[3 * i] = U(r.x); out[3 * i + 1] = U(r.y); out[3 * i + 2] = U(r.z);
            break;
        }
        case 9: {
            float sn, cs;
            sincos_2pi(F(in[i]), sn, cs);
            out[2 * i] = U(sn); out[2 * i + 1] = U(cs);
            break;
        }
        case 10: out[i] = U(atan2_poly(F(in[2 * i]), F(in[2 * i + 1]))); break;
        case 11: out[i] = rng_seed(in[3 * i], in[3 * i + 1], in[3 * i + 2]); break;
        case 12: {  // the division-free n / d and wrap used by k_shade: {n / d, n % d, wrap_index((int)n, (int)d)}
            const uint32_t nn = in[2 * i], dd = in[2 * i + 1];
            const FastDiv f = make_fastdiv(dd);
            const uint32_t q = fast_div(f, nn);
            out[3 * i] = q;
            out[3 * i + 1] = nn - q * dd;
            out[3 * i + 2] = (uint32_t)wrap_index((int)nn, (int)(dd & 0xFFFFu) + 1);
            break;
        }
        case 17: out[i] = octa_encode16(v3(F(in[3 * i]), F(in[3 * i + 1]), F(in[3 * i + 2]))); break;  // shading-record normals
        case 18: {
            const V3 n = octa_decode16(in[i]);
            out[3 * i] = U(n.x);
            out[3 * i + 1] = U(n.y);
            out[3 * i + 2] = U(n.z);
            break;
        }
        default: break;
    }
}
bool selftest_widths(int op, uint32_t* in_w, uint32_t* out_w) {
    static const uint32_t w[19][2] = {{1, 1}, {2, 1}, {2, 1}, {2, 1}, {11, 4}, {4, 11}, {2, 3}, {3, 6}, {3, 3}, {1, 2}, {2, 1}, {3, 1}, {2, 3},
                                      {2, 3}, {3, 9}, {64, 128}, {64, 1}, {3, 1}, {1, 3}};
    if (op < 0 || op > 18) return false;
    *in_w = w[op][0];
    *out_w = w[op][1];
    return true;
}
void launch_selftest(hipStream_t st, int op, const uint32_t* in, uint32_t n, uint32_t* out) {
    if (op >= 13 && op <= 16) return launch_selftest_probes(st, op, in, n, out);
    hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, st, op, in, n, out);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline unsigned grid_for(uint64_t n, unsigned block, unsigned max_blocks) {
    uint64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > max_blocks ? max_blocks : b);
}

void launch_raygen(hipStream_t st, const GConstDev& g, const uint32_t* pixels, uint32_t npix, float* rays, size_t stride) {
    hipLaunchKernelGGL(k_raygen, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, st, g, pixels, npix, rays, stride);
}
void launch_extend(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* rays, size_t stride,
                   const uint32_t* count_ptr, uint32_t count_imm, uint32_t max_n, float* hits, uint32_t* cn, uint32_t* ct,
                   unsigned long long* totals, uint32_t* work_counter, bool payload) {
    unsigned grid = grid_for(max_n, kExtendBlock, g_trace_max_blocks);
#define RT3_LAUNCH_EXTEND(C, L)                                                                                                                  \
    hipLaunchKernelGGL((k_extend<C, L>), dim3(grid), dim3(kExtendBlock), 0, st, nodes, tris, top, n_top, rays, stride, count_ptr, count_imm, hits, cn, ct, totals, \
                       work_counter, payload ? 1 : 0, totals ? totals + 10 : nullptr /* the context's totals block: [10] = LDS-served visits, closest hit */)
    if (count) {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_EXTEND(true, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_EXTEND(true, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_EXTEND(true, kLayoutWide128);
        else RT3_LAUNCH_EXTEND(true, kLayoutBinary64);
    } else {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_EXTEND(false, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_EXTEND(false, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_EXTEND(false, kLayoutWide128);
        else RT3_LAUNCH_EXTEND(false, kLayoutBinary64);
    }
#undef RT3_LAUNCH_EXTEND
}
void launch_shadow(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* rays, size_t stride,
                   const uint32_t* count_ptr, uint32_t count_imm, uint32_t max_n, const float* contrib, const uint32_t* pid, float* lacc,
                   size_t lstride, uint32_t* occluded_out, uint32_t* cn, uint32_t* ct, unsigned long long* totals, uint32_t* work_counter) {
    unsigned grid = grid_for(max_n, kExtendBlock, g_trace_max_blocks);
#define RT3_LAUNCH_SHADOW(C, L)                                                                                                                \
    hipLaunchKernelGGL((k_shadow<C, L>), dim3(grid), dim3(kExtendBlock), 0, st, nodes, tris, top, n_top, rays, stride, count_ptr, count_imm, contrib, pid, lacc, \
                       lstride, occluded_out, cn, ct, totals, work_counter, totals ? totals + 9 : nullptr /* totals = block + 2 here: block[11] = LDS-served visits, any hit */)
    if (count) {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_SHADOW(true, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_SHADOW(true, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_SHADOW(true, kLayoutWide128);
        else RT3_LAUNCH_SHADOW(true, kLayoutBinary64);
    } else {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_SHADOW(false, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_SHADOW(false, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_SHADOW(false, kLayoutWide128);
        else RT3_LAUNCH_SHADOW(false, kLayoutBinary64);
    }
#undef RT3_LAUNCH_SHADOW
}
void launch_trace(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* ext_rays, const float* sh_rays,
                  size_t stride, const uint32_t* ext_count, const uint32_t* sh_count, uint32_t max_n, float* hits, const float* contrib, float* lacc,
                  unsigned long long* totals, uint32_t* work_ext, uint32_t* work_sh) {
    unsigned grid = grid_for(max_n, kExtendBlock, g_trace_max_blocks);
#define RT3_LAUNCH_TRACE(C, L)                                                                                                                  \
    hipLaunchKernelGGL((k_trace<C, L>), dim3(grid), dim3(kExtendBlock), 0, st, nodes, tris, top, n_top, ext_rays, sh_rays, stride, ext_count, sh_count, hits, \
                       contrib, lacc, totals, work_ext, work_sh)
    if (count) {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_TRACE(true, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_TRACE(true, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_TRACE(true, kLayoutWide128);
        else RT3_LAUNCH_TRACE(true, kLayoutBinary64);
    } else {
        if (layout == kLayoutWide48Q) RT3_LAUNCH_TRACE(false, kLayoutWide48Q);
        else if (layout == kLayoutWide64Q) RT3_LAUNCH_TRACE(false, kLayoutWide64Q);
        else if (layout == kLayoutWide128) RT3_LAUNCH_TRACE(false, kLayoutWide128);
        else RT3_LAUNCH_TRACE(false, kLayoutBinary64);
    }
#undef RT3_LAUNCH_TRACE
}
void launch_gbuffer(hipStream_t st, const SceneDev& sc, const uint32_t* pixels, uint32_t npix, uint32_t width, const float* hits,
                    size_t stride, void* gbuffer, float* depth) {
    hipLaunchKernelGGL(k_gbuffer, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, st, sc, pixels, npix, width, hits, stride, (uint4*)gbuffer, depth);
}
__global__ void k_pixbn(const uint32_t* __restrict__ pixels, uint32_t npix, const uint8_t* __restrict__ bluenoise, uint32_t bn_w, uint32_t bn_h,
                        uint2* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const uint32_t xy = pixels[i], px = xy & 0xFFFFu, py = xy >> 16;
        uint32_t bn = 0;
        if (bluenoise) bn = *reinterpret_cast<const uint32_t*>(bluenoise + 4 * ((size_t)(py % bn_h) * bn_w + (px % bn_w)));
        out[i] = make_uint2(xy, bn);
    }
}
void launch_pixbn(hipStream_t st, const uint32_t* pixels, uint32_t npix, const uint8_t* bluenoise, uint32_t bn_w, uint32_t bn_h, uint2* out) {
    hipLaunchKernelGGL(k_pixbn, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, st, pixels, npix, bluenoise, bn_w, bn_h, out);
}
void launch_shade(hipStream_t st, bool first, const ShadeLaunch& L) {
    ShadeArgs a;
    a.g = L.g; a.sc = L.sc; a.pixels = L.pixels; a.npix = L.npix; a.width = L.width; a.s0 = L.s0; a.bounce = L.bounce;
    a.npix_div = make_fastdiv(L.npix);
    a.pixbn = L.pixbn;
    a.gbuffer = (const uint4*)L.gbuffer; a.depth = L.depth;
    a.in_rays = L.in_rays; a.in_hits = L.in_hits; a.in_T = L.in_T; a.in_count = L.in_count; a.n_first = L.n_first;
    a.out_rays = L.out_rays; a.out_T = L.out_T; a.out_count = L.out_count;
    a.sh_rays = L.sh_rays; a.sh_contrib = L.sh_contrib; a.sh_count = L.sh_count;
    a.lacc = L.lacc; a.stride = L.stride;
    unsigned grid = grid_for(L.max_n, kShadeBlock, 8192);
    if (first) hipLaunchKernelGGL((k_shade<true, false>), dim3(grid), dim3(kShadeBlock), 0, st, a);  // the first vertex comes from the G-buffer
    else if (a.sc.shade_geoms != nullptr && a.sc.n_geoms <= kShadeGeomsLds) hipLaunchKernelGGL((k_shade<false, true>), dim3(grid), dim3(kShadeBlock), 0, st, a);
    else hipLaunchKernelGGL((k_shade<false, false>), dim3(grid), dim3(kShadeBlock), 0, st, a);
}
void launch_accumulate(hipStream_t st, const GConstDev& g, const uint32_t* pixels, uint32_t npix, uint32_t width, const float* depth,
                       const float* lacc, size_t stride, uint32_t sb, int first_batch, int last_batch, float* radsum, void* light,
                       const void* prev) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, st, g, pixels, npix, width, depth, lacc, stride, sb,
                       first_batch, last_batch, radsum, (float4*)light, (const float4*)prev);
}
void launch_postprocess(hipStream_t st, const GConstDev& g, const SceneDev& sc, const uint32_t* pixels, uint32_t npix, uint32_t width,
                        const float* depth, const void* in, void* out) {
    hipLaunchKernelGGL(k_postprocess, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, st, g, sc, pixels, npix, width, depth, (const float4*)in,
                       (float4*)out);
}
void launch_pack_tiles(hipStream_t st, const uint32_t* pixels, uint32_t npix, uint32_t width, const void* img, void* dst) {
    hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, st, pixels, npix, width, (const float4*)img, (float4*)dst);
}
void launch_unpack_tiles(hipStream_t st, const uint32_t* pixels, uint32_t npix, uint32_t width, const void* src, void* img) {
    hipLaunchKernelGGL(k_unpack_tiles, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, st, pixels, npix, width, (const float4*)src, (float4*)img);
}

}  // namespace rt3
