// rdr_shadow.hip — classification of the visibility buffer into a covered-pixel list and
// the sun shadow ray of every covered pixel.
//
// The ray query against the Vulkan TLAS/BLAS (assets/shaders/shading.frag.glsl:421-425,
// renderer/src/renderer/acceleration_sturct_utils.rs:23-49,253-279) becomes a software
// two-level BVH traversal.  Whether a ray is blocked decides 65 % of a pixel's brightness,
// so everything here is exact, un-fused IEEE arithmetic, identical to the CPU check.
#include "rdr_pixel.h"

namespace {

// ---- shadow ray -------------------------------------------------------------------
constexpr float kShadowTmin = 1e-2f, kShadowTmax = 1000.0f; // shading.frag.glsl:421-425
// Conservative ray / box slab test on one child of a wide node.  `w` holds one axis of
// the box as an fp16 pair (low half = box minimum, high half = box maximum).  Only "never
// miss a box the ray touches" matters for the result, so the planes are evaluated as
// t = plane * (1/d) + c with c = -o/d precomputed per ray: one mixed-precision FMA per
// plane, with the half -> float conversion folded into it.  The rounding of c, of the fast
// reciprocal and of the FMA is covered by a per-axis margin folded into the two constants
// (c_near = c - m, c_far = c + m).  The pairs are rotated by the ray's direction sign so
// that the low half is always the entry plane.  NaNs (0 * inf, inf - inf) fall out of
// min/max and never reject.
struct SlabRay {
    vec3 inv_d;  // 1 / d
    vec3 c_near; // -o / d - margin
    vec3 c_far;  // -o / d + margin
    uint32_t rx, ry, rz; // 16 where d < 0 on that axis, else 0: rotation that puts the entry plane low
};

RDR_DEV float half_of(uint32_t w, bool hi) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi ? (w >> 16) : (w & 0xFFFFu)));
}

RDR_DEV bool ray_box(uint32_t wx, uint32_t wy, uint32_t wz, const SlabRay& r, float tmin, float tmax, float& t_exit) {
    wx = __builtin_amdgcn_alignbit(wx, wx, r.rx);
    wy = __builtin_amdgcn_alignbit(wy, wy, r.ry);
    wz = __builtin_amdgcn_alignbit(wz, wz, r.rz);
    const float nx = fmaf(half_of(wx, false), r.inv_d.x, r.c_near.x), fx = fmaf(half_of(wx, true), r.inv_d.x, r.c_far.x);
    const float ny = fmaf(half_of(wy, false), r.inv_d.y, r.c_near.y), fy = fmaf(half_of(wy, true), r.inv_d.y, r.c_far.y);
    const float nz = fmaf(half_of(wz, false), r.inv_d.z, r.c_near.z), fz = fmaf(half_of(wz, true), r.inv_d.z, r.c_far.z);
    const float t_in = fmaxf(fmaxf(fmaxf(nx, ny), nz), tmin);
    const float t_out = fminf(fminf(fminf(fx, fy), fz), tmax);
    t_exit = t_out;
    return t_in <= t_out;
}

// Ray / triangle, two-sided, open interval (tmin, tmax), watertight like the ray query it stands for: a ray
// through an edge or a vertex shared by triangles hits at least one of them.  Woop, Benthin and Wald's test
// ("Watertight Ray/Triangle Intersection", 2013): the triangle is moved to the ray's origin and sheared so that
// the ray runs along the axis of its largest direction component; the three edge functions of the projected
// triangle are differences of products of the same two vertices whichever triangle at an edge computes them, so
// their signs agree, and one that comes out exactly zero is recomputed in double precision.  t = T / det is
// compared without the division.  Both faces count, so the paper's swap of the two other axes for a negative
// direction component (it only keeps the winding) is left out.  The same operations in the same order as the
// CPU check.
// The ray's part of it is computed once per ray (Shear, below): the axis, the two shear factors, the
// reciprocal, and the origin with its components already rotated so that the ray's axis comes last.
struct Shear {
    int kz;          // axis of the largest direction component
    float sx, sy, sz; // d[kx] / d[kz], d[ky] / d[kz], 1 / d[kz] with (kx, ky, kz) a cyclic rotation of (x, y, z)
    vec3 o;          // the origin as (o[kx], o[ky], o[kz])
};
RDR_DEV vec3 rotate_axes(vec3 a, int kz) { // (a[kx], a[ky], a[kz])
    return kz == 0 ? mk3(a.y, a.z, a.x) : (kz == 1 ? mk3(a.z, a.x, a.y) : a);
}
RDR_DEV Shear make_shear(vec3 o, vec3 d) {
    Shear s;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    s.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const vec3 r = rotate_axes(d, s.kz);
    s.sx = r.x / r.z; s.sy = r.y / r.z; s.sz = 1.0f / r.z;
    s.o = rotate_axes(o, s.kz);
    return s;
}
RDR_DEV bool ray_triangle(const Shear& s, vec3 p0, vec3 p1, vec3 p2, float tmin, float tmax) {
    const vec3 A = rotate_axes(p0, s.kz) - s.o, B = rotate_axes(p1, s.kz) - s.o, C = rotate_axes(p2, s.kz) - s.o;
    const float Ax = A.x - s.sx * A.z, Ay = A.y - s.sy * A.z;
    const float Bx = B.x - s.sx * B.z, By = B.y - s.sy * B.z;
    const float Cx = C.x - s.sx * C.z, Cy = C.y - s.sy * C.z;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float Az = s.sz * A.z, Bz = s.sz * B.z, Cz = s.sz * C.z;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float a = fabsf(det), Ts = det < 0.0f ? -T : T;
    return Ts > tmin * a && Ts < tmax * a;
}

// Traversal stacks.  The fast one keeps kLdsStack entries per lane in LDS (one
// column per lane: conflict-free) and only records that it overflowed; a ray whose
// traversal overflowed without finding a hit is traced again with the deep,
// scratch-backed stack.  Reasonably balanced trees never reach the second path.
constexpr int kLdsStack = 24;
constexpr int kDeepStack = 160;

struct LdsStack {
    uint32_t* base; // &shared[thread], stride 256 entries, limit + 1 rows
    int sp, limit;
    bool overflow;
    // Branch-free conditional push: the store always happens, into row `limit` (a dump row
    // nobody reads) when there is nothing to push or no room.
    RDR_DEV void push_if(bool p, uint32_t v) {
        const bool go = p & (sp < limit);
        base[(go ? sp : limit) * 256] = v;
        overflow |= p & !go;
        sp += go ? 1 : 0;
    }
    RDR_DEV void push(uint32_t v) { push_if(true, v); }
    RDR_DEV uint32_t pop() { return base[--sp * 256]; }
};

struct DeepStack {
    uint32_t data[kDeepStack + 1];
    int sp;
    bool overflow;
    RDR_DEV void push_if(bool p, uint32_t v) {
        const bool go = p & (sp < kDeepStack);
        data[go ? sp : kDeepStack] = v;
        overflow |= p & !go;
        sp += go ? 1 : 0;
    }
    RDR_DEV void push(uint32_t v) { push_if(true, v); }
    RDR_DEV uint32_t pop() { return data[--sp]; }
};

struct Ray {
    vec3 o, d;
    SlabRay slab;
    Shear shear; // for the triangle tests: only rays in an instance's object space meet triangles
};

// Margin per axis: |c| * 2^-24 (rounding of c) plus (2^-23 + 2^-24) * |t| for the
// reciprocal and the FMA with |t| <= tmax where it can matter; (|c| + tmax) * 2^-21
// covers both four times over.
RDR_DEV Ray make_ray(vec3 o, vec3 d, float tmax) {
    Ray r;
    r.o = o; r.d = d;
    const vec3 inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const vec3 c = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const float k = 4.76837158203125e-7f; // 2^-21
    const vec3 m = mk3((fabsf(c.x) + tmax) * k, (fabsf(c.y) + tmax) * k, (fabsf(c.z) + tmax) * k);
    r.slab.inv_d = inv;
    r.slab.c_near = c - m;
    r.slab.c_far = c + m;
    r.slab.rx = d.x < 0.0f ? 16u : 0u;
    r.slab.ry = d.y < 0.0f ? 16u : 0u;
    r.slab.rz = d.z < 0.0f ? 16u : 0u;
    // the compiler is not to know that these are 0 or 16: it would turn every rotation by them into a rotation by 16
    // and a select on the sign, two instructions (one with a scalar-register mask) for one
    asm volatile("" : "+v"(r.slab.rx), "+v"(r.slab.ry), "+v"(r.slab.rz));
    return r;
}

// The ray in an instance's object space: origin and (unnormalised) direction through
// the inverse model matrix, parameter t unchanged.
RDR_DEV Ray object_ray(const GpuInstance& inst, vec3 o, vec3 d, float tmax) {
    const float* iv = inst.inv;
    const vec3 oo = mk3(((iv[0] * o.x + iv[1] * o.y) + iv[2] * o.z) + iv[3], ((iv[4] * o.x + iv[5] * o.y) + iv[6] * o.z) + iv[7],
                        ((iv[8] * o.x + iv[9] * o.y) + iv[10] * o.z) + iv[11]);
    const vec3 od = mk3((iv[0] * d.x + iv[1] * d.y) + iv[2] * d.z, (iv[4] * d.x + iv[5] * d.y) + iv[6] * d.z,
                        (iv[8] * d.x + iv[9] * d.y) + iv[10] * d.z);
    Ray r = make_ray(oo, od, tmax);
    r.shear = make_shear(oo, od);
    return r;
}

// Squared distance between a ball's centre and the line o + t d, with a bound on what rounding can have moved it:
// `oc` = o - centre, `b` = oc . d, `inv_a` = 1 / (d . d).  The textbook discriminant b^2 - a (|oc|^2 - r^2) is a
// difference of two terms of the size of a |oc|^2 that leaves a r^2: with the origin D = hundreds of radii away
// (a low sun: the ray parameter goes up to 1000) its rounding error, a few 2^-24 a D^2, is larger than the slack the
// radii carry.  Here the component of oc across the ray is formed first, l = oc - (b / a) d, and the discriminant
// is a (r^2 - |l|^2): l's components are wrong by a few 2^-24 |oc|, |l|^2 by 2 |l| times that.
RDR_DEV float line_distance2(vec3 oc, vec3 d, float b, float inv_a, float cc, float& err) {
    const float s = b * inv_a;
    const vec3 l = mk3(oc.x - s * d.x, oc.y - s * d.y, oc.z - s * d.z);
    const float ll = dot3(l, l);
    err = 3.814697265625e-6f * sqrtf(cc * ll) + 1.4551915228366852e-11f * cc; // 2^-18 |oc| |l| + 2^-36 |oc|^2
    return ll;
}

// What an instance's two balls (GpuInstance::outer_radius) say about a ray in its object space (o, d):
//   kBallBlocked  the ray is inside the inner ball somewhere beyond tmin and leaves the outer one before tmax: it is
//                 blocked, no walk needed;
//   kBallMissed   the ray does not come within the outer ball between tmin and tmax: the instance cannot block it;
//   kBallWalk     walk the instance, and nothing beyond t_hi (where the ray leaves the outer ball) can be hit.
// The radii carry their own slack (0.003 % inside, 0.01 % outside), the distance between line and centre carries the
// bound of line_distance2 (a verdict needs the distance to be on its side of the radius by more than that bound;
// otherwise the instance is walked), and the parameters another 1 %.
enum BallVerdict { kBallWalk, kBallBlocked, kBallMissed };
RDR_DEV BallVerdict ball_verdict(float4 ball, float outer_radius, vec3 o, vec3 d, float tmin, float tmax, float& t_hi) {
    t_hi = tmax;
    if (!(outer_radius > 0.0f)) return kBallWalk;
    const vec3 oc = mk3(o.x - ball.x, o.y - ball.y, o.z - ball.z);
    const float a = dot3(d, d), b = dot3(oc, d), cc = dot3(oc, oc);
    if (!(a > 0.0f)) return kBallWalk;
    const float inv_a = 1.0f / a;
    float err;
    const float ll = line_distance2(oc, d, b, inv_a, cc, err);
    const float r2_out = outer_radius * outer_radius;
    if (ll - err > r2_out) return kBallMissed;
    // where the line crosses the outer ball, widened by the bound (so that [t_enter, t_leave] can only grow)
    const float root_out = sqrtf(a * fmaxf((r2_out + err) - ll, 0.0f));
    const float t_leave_out = (root_out - b) * inv_a, t_enter_out = (-root_out - b) * inv_a;
    if (t_leave_out * 1.01f + 1e-30f < tmin || t_enter_out * 0.99f - 1e-30f > tmax) return kBallMissed;
    if (t_leave_out >= 0.0f) t_hi = fminf(tmax, t_leave_out * 1.01f + 1e-30f);
    const float r2_in = ball.w * ball.w;
    if (ball.w > 0.0f && ll + err < r2_in) {
        const float t_leave_in = (sqrtf(a * ((r2_in - err) - ll)) - b) * inv_a; // the inner ball shrunk by the bound
        if (t_leave_in > tmin * 1.01f && t_leave_out < tmax * 0.99f) return kBallBlocked;
    }
    return kBallWalk;
}

// Is anything between the surface point and the sun?  One loop walks both levels: `cur` is
// the child code being processed, the stack holds the ones still to do, and `floor` is the
// stack height at which the current instance's BLAS is exhausted (-1 while in the top
// level).  The result is a plain "any triangle hit", so the visiting order is free: the
// pixel's own instance, by far the likeliest occluder, goes first (with the top level's
// root waiting at the bottom of the stack) and is skipped when the top level meets it
// again.  Every iteration handles one node and, if that left nothing to continue with,
// pops the next; state changes are in place so that the loop has one back edge.
// `skip_own`: the caller has established that the ray cannot meet the pixel's own instance (it leaves
// a closed convex mesh outwards): the walk starts in the state in which that instance's bottom level
// has just been exhausted.
// `mode` says which part of the work this call does (the parts' results OR together, in any order):
//   kTraceAll    everything;
//   kTraceDefer  everything, except that the walk through the own instance is left out, and reported in
//                `deferred`, when that instance has an inner ball (a closed convex mesh): what is left of
//                those after the ball tests are the rays that graze the mesh near its terminator, a few
//                per cent of the rays with a dozen and more steps each through the leaves under the surface,
//                and in a wave of rays that are done after the set-up they would keep the wave alive for
//                their own sake.  k_shadow_walk takes them packed into full waves;
//   kTraceOwn    the walk through the own instance alone.
enum TraceMode : uint32_t { kTraceAll = 0, kTraceDefer = 1, kTraceOwn = 2 };

template <typename Stack>
RDR_DEV bool shadow_any_hit(const SceneView& sv, uint32_t own_instance, bool skip_own, vec3 o, vec3 d, float tmin, float tmax, Stack& st,
                            uint32_t mode, bool& deferred) {
    st.sp = 0;
    st.overflow = false;
    deferred = false;
    bool walk_own = !skip_own;
    float t_hi = tmax; // the walk's far limit inside the current instance
    Ray ray;
    if (walk_own) {
        ray = object_ray(sv.instances[own_instance], o, d, tmax);
        const float4 ball = sv.instance_ball[own_instance];
        const BallVerdict v = ball_verdict(ball, sv.instances[own_instance].outer_radius, ray.o, ray.d, tmin, tmax, t_hi);
        if (v == kBallBlocked) return true;
        if (v == kBallMissed) walk_own = false; // the ray is out of the instance's reach before tmin
        else if (mode == kTraceDefer && ball.w > 0.0f) { walk_own = false; deferred = true; }
    }
    if (mode == kTraceOwn && !walk_own) return false;
    // The world level: the top-level tree from its root, or, where k_sun_candidates left a short list of the instances
    // that can block a sun ray from this one, that list in the form of a top-level leaf (usually empty or one or two
    // instances, against four to six dependent node steps through the tree).
    uint32_t world = 0u; // node 0 of a tree holds the children of its root
    const uint32_t n_cand = sv.sun_cand_count[own_instance];
    if (n_cand <= kSunCandidates)
        world = n_cand ? (kBvhLeaf | (n_cand << 24) | (sv.sun_cand_base + own_instance * kSunCandidates)) : kBvhNone;
    if (mode == kTraceOwn) world = kBvhNone;
    if (!walk_own) {
        if (world == kBvhNone) return false;
        ray = make_ray(o, d, tmax);
        t_hi = tmax;
    } else if (world != kBvhNone) {
        st.push(world);
    }
    int floor = walk_own ? st.sp : -1;
    uint32_t node_base = walk_own ? sv.instances[own_instance].blas_root : sv.tlas_base, tri_base = sv.instances[own_instance].tri_base;
    uint32_t cur = walk_own ? 0u : world;
    bool hit = false;
    for (;;) {
        if (!(cur & kBvhLeaf)) {
            // one wide node: all its loads are in flight together
            const uint4* q = reinterpret_cast<const uint4*>(sv.bvh + (node_base + cur));
            const uint4 codes = q[0], bx = q[1], by = q[2], bz = q[3];
            const uint32_t code[4] = {codes.x, codes.y, codes.z, codes.w};
            const uint32_t wx[4] = {bx.x, bx.y, bx.z, bx.w}, wy[4] = {by.x, by.y, by.z, by.w}, wz[4] = {bz.x, bz.y, bz.z, bz.w};
            cur = kBvhNone;
            // Any hit ends the ray, so the order of the children is free: the one the ray leaves last goes
            // first.  A ray that starts on a closed surface finds its blocker on the far side, and the
            // boxes around its origin, which it cannot hit, are left for last (measured against the stored
            // order: -5 %; nearest first: +7 % on the field, +35 % indoors; a full sort of the four gains nothing more).
            float cur_exit = -1.0e30f;
#pragma unroll
            for (int i = 0; i < kBvhWidth; ++i) {
                float t_exit;
                const bool h = (code[i] != kBvhNone) & ray_box(wx[i], wy[i], wz[i], ray.slab, tmin, t_hi, t_exit);
                const bool take = h & (t_exit >= cur_exit);
                st.push_if(h & (cur != kBvhNone), take ? cur : code[i]);
                cur = take ? code[i] : cur;
                cur_exit = take ? t_exit : cur_exit;
            }
        }
        // not "else": a leaf that the node above just selected is tested in the same iteration
        if ((cur & kBvhLeaf) && cur != kBvhNone) {
            const uint32_t count = (cur >> 24) & 127u, first = cur & 0xFFFFFFu;
            cur = kBvhNone;
            if (floor >= 0) {
                const float* tp = sv.blas_tris + (size_t)(tri_base + first) * 9;
#pragma clang loop vectorize(disable)
                for (uint32_t i = 0; i < count; ++i, tp += 9)
                    hit |= ray_triangle(ray.shear, mk3(tp[0], tp[1], tp[2]), mk3(tp[3], tp[4], tp[5]), mk3(tp[6], tp[7], tp[8]), tmin, t_hi);
            } else {
                // top-level leaf: take its first instance now, leave the others on the stack
                const uint32_t instance = sv.tlas_items[first];
                if (count > 1u) st.push(kBvhLeaf | ((count - 1u) << 24) | (first + 1u));
                // the instance's ball in world space first: a ray that passes it by, or has it behind, needs neither the
                // instance's matrix nor the object-space set-up (three divisions and a dozen loads)
                // (for the candidates only: what the top-level tree hands out has passed its box already, and the extra
                // dependent load in front of every instance costs a scene with many of those more than it saves)
                bool reach = instance != own_instance;
                if (reach && first >= sv.sun_cand_base) {
                    const float4 wb = sv.world_ball[instance];
                    if (wb.w > 0.0f) {
                        const vec3 oc = mk3(o.x - wb.x, o.y - wb.y, o.z - wb.z);
                        const float a = dot3(d, d), b = dot3(oc, d), cc = dot3(oc, oc), r2 = wb.w * wb.w;
                        float err;
                        const float ll = line_distance2(oc, d, b, 1.0f / a, cc, err);
                        // passed by (the line stays outside the ball by more than rounding can account for), or behind
                        // the origin (moving away from a ball the origin is outside of)
                        reach = !(ll - err > r2) && !(b > 0.0f && cc > r2 * 1.0001f);
                    }
                }
                if (reach) {
                    const GpuInstance& inst = sv.instances[instance];
                    ray = object_ray(inst, o, d, tmax);
                    const BallVerdict v = ball_verdict(sv.instance_ball[instance], inst.outer_radius, ray.o, ray.d, tmin, tmax, t_hi);
                    hit = v == kBallBlocked;
                    node_base = inst.blas_root;
                    tri_base = inst.tri_base;
                    floor = st.sp;
                    cur = v == kBallMissed ? kBvhNone : 0u; // a missed instance is exhausted at once
                }
            }
        }
        if (hit) break;
        if (cur == kBvhNone) {
            if (st.sp == floor) { // this instance is done: back to the world ray
                t_hi = tmax;
                floor = -1;
                ray = make_ray(o, d, tmax);
                node_base = sv.tlas_base;
            }
            if (st.sp == 0) break;
            cur = st.pop();
        }
    }
    return hit;
}

// World-space position of the surface point seen through a pixel (the shadow ray's origin).
// `face_normal`: (w2 - w1) x (w3 - w1) of the pixel's triangle in world space, not normalised.
RDR_DEV vec3 pixel_world_position(const SceneView& sv, const rdr_ubo& u, float two_over_width, float two_over_height, const ListedPixel& p, vec3& face_normal) {
    float model[16];
    load_m4(sv.transforms + (size_t)p.model_index * 16, model);
    // positions from the 64-byte vertices (one 12-byte load and one cache line per vertex, the line the shading pass
    // takes next), not from the rasteriser's three planes (three loads, three lines)
    const float* q1 = sv.vertices[p.i1].p;
    const float* q2 = sv.vertices[p.i2].p;
    const float* q3 = sv.vertices[p.i3].p;
    const vec4 w1 = mul_m4_fma(model, mk4(q1[0], q1[1], q1[2], 1.0f));
    const vec4 w2 = mul_m4_fma(model, mk4(q2[0], q2[1], q2[2], 1.0f));
    const vec4 w3 = mul_m4_fma(model, mk4(q3[0], q3[1], q3[2], 1.0f));
    const vec4 c1 = mul_m4_fma(u.view_proj, w1), c2 = mul_m4_fma(u.view_proj, w2), c3 = mul_m4_fma(u.view_proj, w3);
    const BaryDeriv bary = calc_full_bary(c1, c2, c3, p.uvx * 2.0f - 1.0f, p.uvy * 2.0f - 1.0f, two_over_width, two_over_height);
    face_normal = cross3(mk3(w2.x - w1.x, w2.y - w1.y, w2.z - w1.z), mk3(w3.x - w1.x, w3.y - w1.y, w3.z - w1.z));
    return interp3(bary, mk3(w1.x, w1.y, w1.z), mk3(w2.x, w2.y, w2.z), mk3(w3.x, w3.y, w3.z));
}

// Can the pixel's own instance be left out of its shadow ray?  Yes when the instance's mesh is a closed
// convex surface (GpuInstance::convex_cos != 0), the pixel shows a triangle of the very mesh the ray
// structure holds (level of detail 0) and the ray leaves that triangle's plane outwards by more than the
// instance's margin: the whole mesh lies behind the plane, so no triangle of it can be hit beyond tmin.
RDR_DEV bool leaves_convex_mesh(const SceneView& sv, uint32_t draw_id, uint32_t lod, vec3 face_normal, vec3 L) {
    const float c = sv.instances[draw_id].convex_cos;
    if (c == 0.0f || lod != 0u) return false;
    const float out = c > 0.0f ? dot3(face_normal, L) : -dot3(face_normal, L);
    return out > fabsf(c) * len3(face_normal);
}

// How far the 3 x 3 part of a model matrix can stretch a vector, from above: the square root of the largest absolute
// row sum of A^T A (Gershgorin), which for a rotation times a uniform scale is the scale itself.  The longest column
// (what the culling's bounding spheres use, as the reference does) is a lower bound once the matrix shears.
RDR_DEV float model_stretch_bound(const float* m) {
    const vec3 a0 = mk3(m[0], m[1], m[2]), a1 = mk3(m[4], m[5], m[6]), a2 = mk3(m[8], m[9], m[10]);
    const float g00 = dot3(a0, a0), g11 = dot3(a1, a1), g22 = dot3(a2, a2);
    const float g01 = fabsf(dot3(a0, a1)), g02 = fabsf(dot3(a0, a2)), g12 = fabsf(dot3(a1, a2));
    return sqrtf(fmaxf(g00 + g01 + g02, fmaxf(g01 + g11 + g12, g02 + g12 + g22)));
}

// Which instances can block a sun ray that starts on a given instance: one thread per draw sends a fat ray from
// the centre of the instance's ball towards the sun through the top level -- every box widened by the world-space
// radius that holds all of the instance's vertices, which covers every ray that starts on the instance -- and
// collects the other instances in the leaves it reaches.  The boxes bound the instances' geometry, so a triangle of
// an instance that is not collected cannot be hit.  Up to kSunCandidates are kept (more, or a stack overflow: the
// shadow rays walk the top level as before).  Depends on the sun direction and the transforms only: launched
// when either has changed (rdr_shade), not per frame.
// The world-space ball around the mesh the ray structure holds for draw `di` (centre, radius; w = 0: unknown).
RDR_DEV float4 instance_world_ball(const SceneView& sv, uint32_t di, const float* model) {
    const float4 ball = sv.instance_ball[di];
    const vec4 cw = mul_m4(model, mk4(ball.x, ball.y, ball.z, 1.0f));
    const float r0 = sv.instances[di].outer_radius * model_stretch_bound(model) * 1.001f + 1e-5f * (1.0f + fmaxf(fabsf(cw.x), fmaxf(fabsf(cw.y), fabsf(cw.z))));
    return (r0 > 0.0f && r0 < 1e30f) ? make_float4(cw.x, cw.y, cw.z, r0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Can a ray with direction L (unit length) that starts within R of c meet the ball `wb`?  No when the ball's centre is
// farther from the line through c than R plus its radius, or when the whole ball lies behind every such origin; the
// margin covers the rounding of the difference of two far-apart positions.  An unknown ball (w = 0) can.
RDR_DEV bool fat_ray_reaches_ball(vec3 c, vec3 L, float R, float4 wb) {
    if (!(wb.w > 0.0f)) return true;
    const vec3 v = mk3(wb.x - c.x, wb.y - c.y, wb.z - c.z);
    const float t = dot3(v, L);
    const vec3 w = v - L * t;
    const float thr = (R + wb.w) * 1.0001f + 4e-5f * (1.0f + (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z));
    if (dot3(w, w) > thr * thr) return false;
    if (t + thr < 0.0f) return false;
    return true;
}

__global__ __launch_bounds__(256) void k_sun_candidates(SceneView sv, FrameView fv) {
    __shared__ uint32_t stack_mem[(kLdsStack + 1) * 256];
    const uint32_t di = blockIdx.x * 256u + threadIdx.x;
    if (di >= sv.draw_count) return;
    const float reach = sv.instance_reach[di];
    uint32_t count = 255u;
    float4 wball = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (reach > 0.0f) {
        const float4 ball = sv.instance_ball[di];
        float model[16];
        load_m4(sv.transforms + (size_t)sv.draws[di].model_index * 16, model);
        const vec4 cw = mul_m4(model, mk4(ball.x, ball.y, ball.z, 1.0f));
        const vec3 c = mk3(cw.x, cw.y, cw.z);
        // the ball around the mesh the ray structure holds, in world space (for the pre-test in front of an instance's walk)
        wball = instance_world_ball(sv, di, model);
      if (fv.sun_candidates) {
        // world-space radius with room for the rounding of an origin (interpolated world position) and of this sum
        const float R = reach * model_stretch_bound(model) * 1.001f + 1e-4f * (1.0f + fmaxf(fabsf(c.x), fmaxf(fabsf(c.y), fabsf(c.z))));
        const vec3 L = mk3(fv.sun_unit[0], fv.sun_unit[1], fv.sun_unit[2]);
        Ray ray = make_ray(c, L, kShadowTmax + R);
        // widening every box by R is moving the entry planes R / |d| earlier and the exit planes as much later
        const vec3 w = mk3(R * fabsf(ray.slab.inv_d.x), R * fabsf(ray.slab.inv_d.y), R * fabsf(ray.slab.inv_d.z));
        ray.slab.c_near = ray.slab.c_near - w;
        ray.slab.c_far = ray.slab.c_far + w;
        LdsStack st;
        st.base = stack_mem + threadIdx.x;
        st.limit = kLdsStack;
        st.sp = 0;
        st.overflow = false;
        bool give_up = !(R < 1e30f) || !(L.x == L.x); // a radius or a direction that is not a number proves nothing
        uint32_t n = 0, cur = 0u;
        uint32_t* out = sv.sun_cand_items + (size_t)di * kSunCandidates;
        while (!give_up) {
            if (!(cur & kBvhLeaf)) {
                const uint4* q = reinterpret_cast<const uint4*>(sv.bvh + (sv.tlas_base + cur));
                const uint4 codes = q[0], bx = q[1], by = q[2], bz = q[3];
                const uint32_t code[4] = {codes.x, codes.y, codes.z, codes.w};
                const uint32_t wx[4] = {bx.x, bx.y, bx.z, bx.w}, wy[4] = {by.x, by.y, by.z, by.w}, wz[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
                for (int i = 0; i < kBvhWidth; ++i) {
                    float t_exit;
                    st.push_if((code[i] != kBvhNone) & ray_box(wx[i], wy[i], wz[i], ray.slab, -R, kShadowTmax + R, t_exit), code[i]);
                }
            } else {
                const uint32_t items = (cur >> 24) & 127u, first = cur & 0xFFFFFFu;
                for (uint32_t i = 0; i < items; ++i) {
                    const uint32_t other = sv.tlas_items[first + i];
                    if (other == di) continue;
                    // A leaf was reached because its box is; the instances in it are then held to their own balls, which
                    // on a field of like-sized neighbours leaves most lists empty (every pixel of such an instance then
                    // skips the world level of its shadow ray altogether).
                    float other_model[16];
                    load_m4(sv.transforms + (size_t)sv.draws[other].model_index * 16, other_model);
                    if (!fat_ray_reaches_ball(c, L, R, instance_world_ball(sv, other, other_model))) continue;
                    if (n < kSunCandidates) out[n] = other;
                    ++n;
                }
            }
            if (st.sp == 0 || n > kSunCandidates) break;
            cur = st.pop();
        }
        // (a top-level leaf code holds its first item in 24 bits)
        if (!give_up && !st.overflow && n <= kSunCandidates && sv.sun_cand_base + (di + 1u) * kSunCandidates < (1u << 24)) count = n;
      }
    }
    sv.sun_cand_count[di] = (uint8_t)count;
    sv.world_ball[di] = wball;
}

// Pass 0: classification.  A workgroup of sixteen waves owns a 128x128-pixel region as four 64x64
// quadrants; a quadrant is sixteen 16x16 blocks, and each of its four waves takes an 8x8 sub-block of
// every block.  Pixels with no geometry get the shading pass's clear colour here; covered pixels are
// appended to a compact pixel list, which the two passes below run over with full waves.  One atomic
// per workgroup reserves the region's span of the list: atomics that return a value serialise on their
// address at about 10 ns each, so their number per frame is what bounds this kernel (64x64 regions,
// 2040 atomics at 4K: 0.031 ms; 128x128: 510).  Inside the span the order is quadrant-, block-, wave-,
// lane-major, so 64 consecutive entries stay spatially close.
constexpr uint32_t kClassifyRegion = 128;

__global__ __launch_bounds__(1024) void k_classify(FrameView fv) {
    __shared__ uint32_t wave_count[4 * 16 * 4]; // [quadrant][block][wave of the quadrant]
    __shared__ uint32_t scan_total[4];
    __shared__ uint32_t wg_base;
    const uint32_t lane = threadIdx.x & 63u, wave = (threadIdx.x >> 6) & 3u, quad = threadIdx.x >> 8;
    const uint32_t x0 = blockIdx.x * kClassifyRegion + (quad & 1u) * 64 + (wave & 1u) * 8 + (lane & 7u);
    const uint32_t v0 = blockIdx.y * kClassifyRegion + (quad >> 1) * 64 + (wave >> 1) * 8 + (lane >> 3); // of the rows this renderer shades
    uint32_t covered_bits = 0;
    uint32_t ids[16]; // low words of the covered pixels' keys, for the list
    uint32_t* color = reinterpret_cast<uint32_t*>(fv.color);
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const uint32_t ix = x0 + (b & 3u) * 16, vy = v0 + (b >> 2) * 16;
        bool covered = false;
        if (ix < fv.width && vy < fv.shade_rows) {
            const uint32_t iy = rdr_shade_row(fv, vy);
            const uint32_t pi = iy * fv.width + ix;
            const unsigned long long key = fv.vis[pi];
            covered = key != RDR_VIS_CLEAR;
            ids[b] = (uint32_t)key;
            if (!covered) color[pi] = 0u; // the shading pass clears its target to (0,0,0,0)
        }
        const unsigned long long m = __ballot(covered);
        if (lane == 0) wave_count[quad * 64 + b * 4 + wave] = (uint32_t)__popcll(m);
        covered_bits |= covered ? (1u << b) : 0u;
    }
    __syncthreads();
    // exclusive prefix sum of the 256 (quadrant, block, wave) counts: a wave per quadrant, then the four totals
    uint32_t c = 0, inc = 0;
    if (threadIdx.x < 256) {
        c = wave_count[threadIdx.x];
        inc = wave_scan_inclusive(c, lane);
        if (lane == 63) scan_total[threadIdx.x >> 6] = inc;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const uint32_t q = threadIdx.x >> 6;
        const uint32_t t0 = scan_total[0], t1 = scan_total[1], t2 = scan_total[2], t3 = scan_total[3];
        const uint32_t before = (q > 0 ? t0 : 0u) + (q > 1 ? t1 : 0u) + (q > 2 ? t2 : 0u);
        wave_count[threadIdx.x] = before + inc - c;
        if (threadIdx.x == 0) {
            const uint32_t total = t0 + t1 + t2 + t3;
            wg_base = total ? atomicAdd(&fv.counters->pixel_count, total) : 0u;
        }
    }
    __syncthreads();
    const uint32_t base = wg_base;
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const bool covered = (covered_bits >> b) & 1u;
        const unsigned long long m = __ballot(covered);
        if (covered) {
            const uint32_t ix = x0 + (b & 3u) * 16, iy = rdr_shade_row(fv, v0 + (b >> 2) * 16);
            const uint32_t at = base + wave_count[quad * 64 + b * 4 + wave] + ballot_rank(m);
            fv.pixel_list[at] = (iy << 16) | ix;
            fv.pixel_key[at] = ids[b];
        }
    }
}

// Pass 1: the sun shadow ray of every covered pixel (shading.frag.glsl:421-425), kept
// apart from the material evaluation so that the pointer-chasing traversal runs at
// high occupancy with its stack in LDS.  The result is stored per list entry:
// 0 lit, 1 blocked, and two codes that k_shadow_walk resolves to 0 or 1.
constexpr uint8_t kShadowLit = 0, kShadowBlocked = 1, kShadowRetrace = 2, kShadowWalkOwn = 3;

struct ShadowRay {
    uint32_t draw_id;
    bool skip_own;
    vec3 o, d;
};
RDR_DEV ShadowRay shadow_ray_of(const SceneView& sv, const FrameView& fv, uint32_t entry) {
    const ListedPixel p = listed_pixel(sv, fv, entry);
    vec3 face_normal;
    ShadowRay r;
    r.o = pixel_world_position(sv, fv.ubo, fv.two_over_width, fv.two_over_height, p, face_normal);
    r.d = mk3(fv.sun_unit[0], fv.sun_unit[1], fv.sun_unit[2]);
    r.draw_id = p.draw_id;
    r.skip_own = (fv.convex_skip & 1u) && leaves_convex_mesh(sv, p.draw_id, p.lod, face_normal, r.d);
    return r;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_shadow(SceneView sv, FrameView fv) {
    __shared__ uint32_t stack_mem[(kLdsStack + 1) * 256];
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry >= fv.counters->pixel_count) return;
    const ShadowRay r = shadow_ray_of(sv, fv, entry);
    LdsStack st;
    st.base = stack_mem + threadIdx.x;
    st.limit = (int)fv.lds_stack_limit;
    bool deferred;
    const bool h = shadow_any_hit(sv, r.draw_id, r.skip_own, r.o, r.d, kShadowTmin, kShadowTmax, st, fv.shadow_defer ? kTraceDefer : kTraceAll, deferred);
    uint8_t code = h ? kShadowBlocked : (deferred ? kShadowWalkOwn : kShadowLit);
    if (!h && st.overflow) { // a miss that dropped stack entries proves nothing: traced again with the deep stack
        code = kShadowRetrace;
        atomicAdd(&fv.counters->shadow_overflow, 1u);
    }
    fv.shadow[entry] = code;
}

// Pass 1b: the rays k_shadow left open.  A workgroup reads the result bytes of kWalkChunk consecutive list
// entries (sixteen per thread), collects the entries with an open code in LDS and traces those with full
// waves: the deferred walks through the ray's own closed convex instance with the LDS stack, and, with the
// deep scratch-backed stack, whatever overflowed the LDS stack (in k_shadow or here).  The two kinds are
// rare and evenly spread over the list, so every workgroup has about the same little to do.
constexpr uint32_t kWalkChunk = 4096;
// The LDS a workgroup holds decides how many of them walk on a compute unit at a time (each has one full wave of rays
// and a second with the rest): sixteen stack rows (the walks through one instance of the scenes measured never hold
// more than eight entries; more than sixteen go to the deep stack) and 16-bit list entries make it 25 KB, six
// workgroups per compute unit, where the full 24 rows and 32-bit entries made it 42 KB and three -- an 8K frame's
// 4 700 workgroups then went through in six rounds.
constexpr int kWalkStackRows = 16;
static_assert(kWalkStackRows <= kLdsStack && kWalkChunk <= (1u << 14), "a list entry is the offset in the chunk and a two-bit code in 16 bits");

__global__ __launch_bounds__(256) void k_shadow_walk(SceneView sv, FrameView fv) {
    __shared__ uint32_t stack_mem[(kWalkStackRows + 1) * 256];
    __shared__ uint16_t open_list[kWalkChunk];
    __shared__ uint32_t open_count;
    const uint32_t count = fv.counters->pixel_count;
    const uint32_t chunk0 = blockIdx.x * kWalkChunk;
    if (chunk0 >= count) return;
    if (threadIdx.x == 0) open_count = 0u;
    __syncthreads();
    const uint32_t e0 = chunk0 + threadIdx.x * 16u;
    if (e0 < count) { // the buffer is padded to a multiple of 16 bytes; entries past the list's end hold older frames' bytes
        const uint4 q = *reinterpret_cast<const uint4*>(fv.shadow + e0);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t code = (w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu;
            if (code >= kShadowRetrace && e0 + k < count) open_list[atomicAdd(&open_count, 1u)] = (uint16_t)(((threadIdx.x * 16u + k) << 2) | code);
        }
    }
    __syncthreads();
    const uint32_t n = open_count;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const uint32_t entry = chunk0 + (open_list[i] >> 2);
        uint32_t code = open_list[i] & 3u;
        const ShadowRay r = shadow_ray_of(sv, fv, entry);
        bool deferred, h = false;
        if (code == kShadowWalkOwn) {
            LdsStack st;
            st.base = stack_mem + threadIdx.x;
            st.limit = min((int)fv.lds_stack_limit, kWalkStackRows);
            h = shadow_any_hit(sv, r.draw_id, r.skip_own, r.o, r.d, kShadowTmin, kShadowTmax, st, kTraceOwn, deferred);
            code = (!h && st.overflow) ? kShadowRetrace + 4u : 0u; // 6: the own instance once more, with the deep stack
        }
        if (code != 0u) {
            DeepStack st;
            h = shadow_any_hit(sv, r.draw_id, r.skip_own, r.o, r.d, kShadowTmin, kShadowTmax, st, code == kShadowRetrace ? kTraceAll : kTraceOwn, deferred);
            if (!h && st.overflow) atomicAdd(&fv.counters->deep_overflow, 1u); // a miss that dropped entries proves nothing: reported
        }
        fv.shadow[entry] = h ? kShadowBlocked : kShadowLit;
    }
}

} // namespace

int rdr_launch_sun_candidates(const SceneView& sv, const FrameView& fv, hipStream_t stream) {
    if (sv.draw_count == 0) return 0;
    hipLaunchKernelGGL(k_sun_candidates, dim3((sv.draw_count + 255u) / 256u), dim3(256), 0, stream, sv, fv);
    RDR_HIP(hipGetLastError());
    return 0;
}

int rdr_launch_shadow(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev) {
    if (fv.shade_rows == 0u) return 0;
    const dim3 tiles((fv.width + kClassifyRegion - 1u) / kClassifyRegion, (fv.shade_rows + kClassifyRegion - 1u) / kClassifyRegion);
    // the list passes are launched for the worst case (every pixel covered); workgroups
    // beyond the list's length exit on their first instruction
    const uint32_t list_blocks = (uint32_t)(((size_t)fv.shade_rows * fv.width + 255u) / 256u);
    if (ev) RDR_HIP(hipEventRecord(ev[6], stream));
    // pixel_count, shadow_overflow: zeroed by the geometry pass's clear; only a second shading pass over the
    // same visibility buffer has to reset them (a memset between kernels is a 4 us kernel of its own)
    if (!fv.list_heads_zero) RDR_HIP(hipMemsetAsync(&fv.counters->pixel_count, 0, 2 * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_classify, tiles, dim3(1024), 0, stream, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[7], stream));
    hipLaunchKernelGGL(k_shadow, dim3(list_blocks), dim3(256), 0, stream, sv, fv);
    hipLaunchKernelGGL(k_shadow_walk, dim3((list_blocks * 256u + kWalkChunk - 1u) / kWalkChunk), dim3(256), 0, stream, sv, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[8], stream));
    RDR_HIP(hipGetLastError());
    return 0;
}
