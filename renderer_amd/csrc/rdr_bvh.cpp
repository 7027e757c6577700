// rdr_bvh.cpp — host-side build of the shadow-ray acceleration structure.
//
// Stands in for the Vulkan BLAS/TLAS the reference builds at scene load
// (renderer/src/renderer/acceleration_sturct_utils.rs:15-49: one BLAS per mesh
// over its LOD0 index list; :253-310: one TLAS instance per draw carrying the
// draw's model matrix; everything flagged opaque).  Two-level binary BVH with
// binned-SAH splits; leaves hold up to four triangles stored as bare positions
// in leaf order so the traversal kernel reads them contiguously.
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include <mutex>
#include <unordered_map>

#include "rdr_internal.h"

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box& b) {
        grow(b.lo);
        grow(b.hi);
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Item {
    Box box;
    float c[3];
    uint32_t index;
};

constexpr int kBins = 12;
constexpr int kLeafSize = 1; // triangle tests are the expensive step of a traversal: one per leaf measured fastest

// Recursive binned-SAH build over items[first, first+count). Node indices are
// relative to `nodes` (the caller offsets them), children adjacent.
void build_rec(std::vector<BvhNode>& nodes, std::vector<Item>& items, uint32_t node, uint32_t first, uint32_t count,
               int depth) {
    Box bounds, cbounds;
    bounds.reset();
    cbounds.reset();
    for (uint32_t i = 0; i < count; ++i) {
        bounds.grow(items[first + i].box);
        cbounds.grow(items[first + i].c);
    }
    for (int k = 0; k < 3; ++k) {
        nodes[node].lo[k] = bounds.lo[k];
        nodes[node].hi[k] = bounds.hi[k];
    }
    auto make_leaf = [&]() {
        nodes[node].left = (int32_t)first;
        nodes[node].count = (int32_t)count;
    };
    if (count <= (uint32_t)kLeafSize || (depth > 48 && count <= kBvhMaxLeaf)) {
        make_leaf();
        return;
    }
    int axis = 0;
    float ext = cbounds.hi[0] - cbounds.lo[0];
    for (int k = 1; k < 3; ++k)
        if (cbounds.hi[k] - cbounds.lo[k] > ext) { ext = cbounds.hi[k] - cbounds.lo[k]; axis = k; }
    uint32_t mid = count / 2;
    bool split_done = false;
    if (ext > 0.0f && count > 8) {
        Box bin_box[kBins];
        uint32_t bin_n[kBins] = {0};
        for (auto& b : bin_box) b.reset();
        const float scale = (float)kBins / ext;
        for (uint32_t i = 0; i < count; ++i) {
            int b = (int)((items[first + i].c[axis] - cbounds.lo[axis]) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            bin_n[b]++;
            bin_box[b].grow(items[first + i].box);
        }
        float right_area[kBins];
        uint32_t right_n[kBins];
        Box acc;
        acc.reset();
        uint32_t n = 0;
        for (int b = kBins - 1; b > 0; --b) {
            if (bin_n[b]) acc.grow(bin_box[b]);
            n += bin_n[b];
            right_area[b] = acc.half_area();
            right_n[b] = n;
        }
        acc.reset();
        n = 0;
        float best = std::numeric_limits<float>::infinity();
        int best_b = -1;
        for (int b = 0; b < kBins - 1; ++b) {
            if (bin_n[b]) acc.grow(bin_box[b]);
            n += bin_n[b];
            if (n == 0 || right_n[b + 1] == 0) continue;
            const float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_n[b + 1];
            if (cost < best) { best = cost; best_b = b; }
        }
        if (best_b >= 0) {
            const float lo = cbounds.lo[axis];
            auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](const Item& it_) {
                int b = (int)((it_.c[axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_b;
            });
            mid = (uint32_t)(it - (items.begin() + first));
            split_done = mid > 0 && mid < count;
        }
    }
    if (!split_done) {
        mid = count / 2;
        std::nth_element(items.begin() + first, items.begin() + first + mid, items.begin() + first + count,
                         [axis](const Item& a, const Item& b) { return a.c[axis] < b.c[axis]; });
    }
    const uint32_t l = (uint32_t)nodes.size();
    nodes.emplace_back();
    nodes.emplace_back();
    nodes[node].left = (int32_t)l;
    nodes[node].count = 0;
    build_rec(nodes, items, l, first, mid, depth + 1);
    build_rec(nodes, items, l + 1, first + mid, count - mid, depth + 1);
}

void build(std::vector<BvhNode>& nodes, std::vector<Item>& items) {
    nodes.clear();
    nodes.emplace_back();
    if (items.empty()) {
        BvhNode& r = nodes[0];
        for (int k = 0; k < 3; ++k) { r.lo[k] = 1.0f; r.hi[k] = -1.0f; } // never hit
        r.left = 0;
        r.count = 0;
        return;
    }
    build_rec(nodes, items, 0, 0, (uint32_t)items.size(), 0);
}

// float -> half with directed rounding (down: largest half <= x, up: smallest half >= x).
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 31u, man = h & 1023u;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { // subnormal
            int e = -1;
            uint32_t m = man;
            do { m <<= 1; ++e; } while (!(m & 1024u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 1023u) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}
uint16_t float_to_half_dir(float x, bool up) {
    if (std::isnan(x)) return up ? 0x7C00u : 0xFC00u; // unbounded side
    // candidates ordered as signed magnitudes: search the monotone sequence by bisection
    // over the 16-bit ordered key (0x0000..0x7C00 positive, mirrored negative)
    auto key_to_half = [](int k) -> uint16_t { return k >= 0 ? (uint16_t)k : (uint16_t)(0x8000u | (uint32_t)(-k)); };
    int lo = -0x7C00, hi = 0x7C00; // -inf .. +inf
    if (up) { // smallest k with value >= x
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (half_to_float(key_to_half(mid)) >= x) hi = mid; else lo = mid + 1;
        }
    } else { // largest k with value <= x
        while (lo < hi) {
            const int mid = lo + (hi - lo + 1) / 2;
            if (half_to_float(key_to_half(mid)) <= x) lo = mid; else hi = mid - 1;
        }
    }
    return key_to_half(lo);
}

uint32_t leaf_code(const BvhNode& n) { return kBvhLeaf | ((uint32_t)n.count << 24) | (uint32_t)n.left; }

float node_half_area(const BvhNode& n) {
    const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Collapse the binary tree into kBvhWidth-wide nodes (appended to `out`, indices relative to
// `base`): starting from a node's two children, the inner child with the largest surface is
// replaced by its own two children until the node is full.
uint32_t emit_wide(const std::vector<BvhNode>& nodes, uint32_t node, std::vector<BvhWide>& out, size_t base) {
    const uint32_t self = (uint32_t)(out.size() - base);
    out.emplace_back();
    uint32_t slots[kBvhWidth];
    int n = 2;
    slots[0] = (uint32_t)nodes[node].left;
    slots[1] = (uint32_t)nodes[node].left + 1;
    while (n < kBvhWidth) {
        int best = -1;
        float best_area = -1.0f;
        for (int i = 0; i < n; ++i) {
            if (nodes[slots[i]].count != 0) continue;
            const float a = node_half_area(nodes[slots[i]]);
            if (a > best_area) { best_area = a; best = i; }
        }
        if (best < 0) break;
        const uint32_t l = (uint32_t)nodes[slots[best]].left;
        slots[best] = l;
        slots[n++] = l + 1;
    }
    uint32_t codes[kBvhWidth];
    for (int i = 0; i < kBvhWidth; ++i) {
        if (i >= n) { codes[i] = kBvhNone; continue; }
        const BvhNode& c = nodes[slots[i]];
        codes[i] = c.count ? leaf_code(c) : emit_wide(nodes, slots[i], out, base);
    }
    BvhWide& w = out[base + self];
    for (int i = 0; i < kBvhWidth; ++i) {
        w.child[i] = codes[i];
        for (int k = 0; k < 3; ++k) {
            w.box[k][i][0] = i < n ? float_to_half_dir(nodes[slots[i]].lo[k], false) : (uint16_t)0;
            w.box[k][i][1] = i < n ? float_to_half_dir(nodes[slots[i]].hi[k], true) : (uint16_t)0;
        }
    }
    return self;
}

void flatten(const std::vector<BvhNode>& nodes, std::vector<BvhWide>& out) {
    const size_t base = out.size();
    const BvhNode& root = nodes[0];
    if (root.count == 0 && root.left != 0) { emit_wide(nodes, 0, out, base); return; }
    BvhWide w{};
    for (int i = 0; i < kBvhWidth; ++i) w.child[i] = kBvhNone;
    if (root.count) {
        w.child[0] = leaf_code(root);
        for (int k = 0; k < 3; ++k) {
            w.box[k][0][0] = float_to_half_dir(root.lo[k], false);
            w.box[k][0][1] = float_to_half_dir(root.hi[k], true);
        }
    }
    out.push_back(w);
}

uint32_t mesh_index_count(const rdr_scene_desc& s, uint32_t mi) {
    const uint32_t end = mi + 1 < s.mesh_count ? s.meshes[mi + 1].index_offset : s.index_count;
    return end - s.meshes[mi].index_offset;
}

// Inverse of the affine part of a column-major mat4: adjugate over determinant
// in double precision, every element rounded to float once.
void affine_inverse(const float* m, float* out12) {
    const double a00 = m[0], a10 = m[1], a20 = m[2];
    const double a01 = m[4], a11 = m[5], a21 = m[6];
    const double a02 = m[8], a12 = m[9], a22 = m[10];
    const double tx = m[12], ty = m[13], tz = m[14];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    const double id = 1.0 / det;
    const double i00 = c00 * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
    const double i10 = c01 * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
    const double i20 = c02 * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
    out12[0] = (float)i00; out12[1] = (float)i01; out12[2] = (float)i02;
    out12[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out12[4] = (float)i10; out12[5] = (float)i11; out12[6] = (float)i12;
    out12[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out12[8] = (float)i20; out12[9] = (float)i21; out12[10] = (float)i22;
    out12[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

// Is the mesh's LOD-0 triangle list the closed, consistently wound boundary of a convex body?
// Vertices are welded by position first (uv seams duplicate them).  Closed and consistently wound: every
// directed edge occurs once and so does its reverse.  Convex: every vertex lies on or behind every face
// plane, up to `tol`; what is tolerated is returned as the protrusion, which the ray margin then covers.
// Quadratic in the mesh size, so only meshes up to a work limit are examined (larger ones count as not
// convex), and results are remembered per mesh content for the life of the process.
struct Convexity {
    int8_t sign = 0;
    float protrusion = 0.0f;
    // a ball inside the body and one around it, about the centre of its bounding box: every face plane at least
    // r_in away from the centre (0 when the centre is not inside), every vertex within r_out
    float centre[3] = {0.0f, 0.0f, 0.0f};
    float r_in = 0.0f, r_out = 0.0f;
};

uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

Convexity analyse_convexity(const rdr_scene_desc& s, uint32_t mi) {
    const rdr_mesh& mesh = s.meshes[mi];
    const uint32_t nidx = mesh_index_count(s, mi), ntri = nidx / 3;
    Convexity none;
    if (ntri < 4) return none;
    // weld by position
    std::vector<uint32_t> weld(mesh.vertex_count);
    std::vector<double> px, py, pz;
    {
        struct Key {
            uint32_t x, y, z;
            bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
        };
        struct Hash {
            size_t operator()(const Key& k) const { return (size_t)(((uint64_t)k.x * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)k.y << 21) ^ ((uint64_t)k.z * 0xC2B2AE3D27D4EB4Full)); }
        };
        std::unordered_map<Key, uint32_t, Hash> ids;
        ids.reserve(mesh.vertex_count * 2);
        for (uint32_t v = 0; v < mesh.vertex_count; ++v) {
            const float* p = s.vertices[mesh.vertex_offset + v].p;
            Key k;
            float q[3] = {p[0] + 0.0f, p[1] + 0.0f, p[2] + 0.0f}; // -0 and +0 are the same place
            std::memcpy(&k, q, 12);
            auto it = ids.find(k);
            if (it == ids.end()) {
                it = ids.emplace(k, (uint32_t)px.size()).first;
                px.push_back(p[0]); py.push_back(p[1]); pz.push_back(p[2]);
            }
            weld[v] = it->second;
        }
    }
    const size_t nv = px.size();
    if ((double)ntri * (double)nv > 1.0e9) return none;
    std::vector<uint32_t> tri(3 * (size_t)ntri);
    std::unordered_map<uint64_t, uint32_t> edges;
    edges.reserve(3 * (size_t)ntri * 2);
    for (uint32_t t = 0; t < ntri; ++t) {
        uint32_t v[3];
        for (int k = 0; k < 3; ++k) v[k] = tri[3 * (size_t)t + k] = weld[s.indices[mesh.index_offset + 3 * t + k]];
        if (v[0] == v[1] || v[1] == v[2] || v[2] == v[0]) return none;
        for (int k = 0; k < 3; ++k)
            if (++edges[((uint64_t)v[k] << 32) | v[(k + 1) % 3]] > 1) return none;
    }
    for (const auto& e : edges) {
        const auto r = edges.find((e.first << 32) | (e.first >> 32));
        if (r == edges.end()) return none;
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t v = 0; v < nv; ++v) {
        lo[0] = std::min(lo[0], px[v]); hi[0] = std::max(hi[0], px[v]);
        lo[1] = std::min(lo[1], py[v]); hi[1] = std::max(hi[1], py[v]);
        lo[2] = std::min(lo[2], pz[v]); hi[2] = std::max(hi[2], pz[v]);
    }
    const double extent = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    const double tol = 1e-5 * extent;
    const double ctr[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
    double r_out = 0.0, face_near = 1e300, face_far = -1e300; // signed distances of the face planes from the centre
    for (size_t v = 0; v < nv; ++v)
        r_out = std::max(r_out, std::sqrt((px[v] - ctr[0]) * (px[v] - ctr[0]) + (py[v] - ctr[1]) * (py[v] - ctr[1]) + (pz[v] - ctr[2]) * (pz[v] - ctr[2])));
    double above = 0.0, below = 0.0; // largest distance of a vertex in front of / behind a face plane
    for (uint32_t t = 0; t < ntri; ++t) {
        const uint32_t a = tri[3 * (size_t)t], b = tri[3 * (size_t)t + 1], c = tri[3 * (size_t)t + 2];
        const double e1[3] = {px[b] - px[a], py[b] - py[a], pz[b] - pz[a]}, e2[3] = {px[c] - px[a], py[c] - py[a], pz[c] - pz[a]};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(len > 0.0)) return none;
        n[0] /= len; n[1] /= len; n[2] /= len;
        const double d0 = n[0] * px[a] + n[1] * py[a] + n[2] * pz[a];
        const double dc = d0 - (n[0] * ctr[0] + n[1] * ctr[1] + n[2] * ctr[2]);
        face_near = std::min(face_near, dc);
        face_far = std::max(face_far, dc);
        double mx = -1e300, mn = 1e300;
        for (size_t v = 0; v < nv; ++v) {
            const double d = n[0] * px[v] + n[1] * py[v] + n[2] * pz[v];
            mx = std::max(mx, d);
            mn = std::min(mn, d);
        }
        above = std::max(above, mx - d0);
        below = std::max(below, d0 - mn);
        if (above > tol && below > tol) return none;
    }
    Convexity out;
    if (above <= tol) { out.sign = 1; out.protrusion = (float)above; }
    else { out.sign = -1; out.protrusion = (float)below; }
    for (int k = 0; k < 3; ++k) out.centre[k] = (float)ctr[k];
    out.r_out = (float)(r_out * 1.0001);
    // normals point outwards for sign +1 (every plane in front of the centre), inwards for -1; the deviation from
    // exact convexity is taken off
    const double r_in = (out.sign > 0 ? face_near : -face_far) - 2.0 * (double)out.protrusion - tol;
    out.r_in = r_in > 0.0 ? (float)r_in : 0.0f;
    return out;
}

Convexity mesh_convexity(const rdr_scene_desc& s, uint32_t mi) {
    static std::mutex mutex;
    static std::unordered_map<uint64_t, std::pair<uint64_t, Convexity>> cache;
    const rdr_mesh& mesh = s.meshes[mi];
    const uint32_t nidx = mesh_index_count(s, mi);
    uint64_t h1 = 0xCBF29CE484222325ull, h2 = 0x84222325CBF29CE4ull;
    for (uint32_t v = 0; v < mesh.vertex_count; ++v) {
        h1 = fnv1a(s.vertices[mesh.vertex_offset + v].p, 12, h1);
        h2 = fnv1a(s.vertices[mesh.vertex_offset + v].p, 12, h2 * 31 + 7);
    }
    h1 = fnv1a(s.indices + mesh.index_offset, (size_t)nidx * 4, h1);
    h2 = fnv1a(s.indices + mesh.index_offset, (size_t)nidx * 4, h2 * 31 + 7);
    {
        std::lock_guard<std::mutex> lock(mutex);
        const auto it = cache.find(h1);
        if (it != cache.end() && it->second.first == h2) return it->second.second;
    }
    const Convexity c = analyse_convexity(s, mi);
    std::lock_guard<std::mutex> lock(mutex);
    cache[h1] = {h2, c};
    return c;
}

} // namespace

void rdr_build_blas(const rdr_scene_desc& s, HostBvh& out) {
    out.blas_nodes.clear();
    out.blas_pairs.clear();
    out.mesh_bounds.assign(s.mesh_count, BvhNode{});
    out.blas_tris.clear();
    out.mesh_root.assign(s.mesh_count, 0);
    out.mesh_tri_base.assign(s.mesh_count, 0);
    out.mesh_convex_sign.assign(s.mesh_count, 0);
    out.mesh_protrusion.assign(s.mesh_count, 0.0f);
    out.mesh_ball.assign(5 * (size_t)s.mesh_count, 0.0f);
    out.mesh_reach.assign(s.mesh_count, 0.0f);
    std::vector<BvhNode> nodes;
    std::vector<Item> items;
    for (uint32_t mi = 0; mi < s.mesh_count; ++mi) {
        const rdr_mesh& mesh = s.meshes[mi];
        const uint32_t ntri = mesh_index_count(s, mi) / 3;
        items.resize(ntri);
        for (uint32_t t = 0; t < ntri; ++t) {
            Item& it = items[t];
            it.box.reset();
            for (int k = 0; k < 3; ++k) {
                const uint32_t vi = s.indices[mesh.index_offset + 3 * t + k] + mesh.vertex_offset;
                it.box.grow(s.vertices[vi].p);
            }
            for (int k = 0; k < 3; ++k) {
                // keep flat triangles from producing zero-thickness slabs
                const float pad = 2e-6f * std::max(std::fabs(it.box.lo[k]), std::fabs(it.box.hi[k])) + 1e-30f;
                it.box.lo[k] -= pad;
                it.box.hi[k] += pad;
                it.c[k] = 0.5f * (it.box.lo[k] + it.box.hi[k]);
            }
            it.index = t;
        }
        build(nodes, items);
        {
            const Convexity cx = mesh_convexity(s, mi);
            out.mesh_convex_sign[mi] = cx.sign;
            out.mesh_protrusion[mi] = cx.protrusion;
            // every mesh: a ball around it (centre of the root box, farthest LOD-0 vertex, 0.01 % slack); closed convex
            // ones: a ball inside it as well (see GpuInstance::outer_radius)
            const BvhNode& rb = nodes[0];
            if (ntri && rb.lo[0] <= rb.hi[0]) {
                double c[3], far2 = 0.0;
                for (int k = 0; k < 3; ++k) c[k] = 0.5 * ((double)rb.lo[k] + (double)rb.hi[k]);
                if (cx.sign != 0 && cx.r_in > 0.0f)
                    for (int k = 0; k < 3; ++k) c[k] = cx.centre[k];
                for (uint32_t i = 0; i < 3 * ntri; ++i) {
                    const float* p = s.vertices[s.indices[mesh.index_offset + i] + mesh.vertex_offset].p;
                    far2 = std::max(far2, (p[0] - c[0]) * (p[0] - c[0]) + (p[1] - c[1]) * (p[1] - c[1]) + (p[2] - c[2]) * (p[2] - c[2]));
                }
                for (int k = 0; k < 3; ++k) out.mesh_ball[5 * (size_t)mi + k] = (float)c[k];
                // inner radius: what the ball test's own rounding can move (a few units in the last place of the
                // largest coordinate it handles, and the centre's rounding to single precision) taken off thirty
                // times over; every 0.01 % given away here widens the band of grazing rays that have to be walked
                double cmax = 0.0;
                for (int k = 0; k < 3; ++k) cmax = std::max(cmax, std::fabs(c[k]));
                const double r_in = (cx.sign != 0 && cx.r_in > 0.0f) ? (double)cx.r_in * (1.0 - 3e-5) - 2.4e-7 * (cmax + std::sqrt(far2)) : 0.0;
                out.mesh_ball[5 * (size_t)mi + 3] = r_in > 0.0 ? (float)r_in : 0.0f;
                out.mesh_ball[5 * (size_t)mi + 4] = (float)(std::sqrt(far2) * 1.0001 + 1e-30);
                // and the distance that holds every vertex of the mesh, whichever level of detail uses it: a pixel may show a
                // coarser level than the ray structure holds, and its shadow ray starts there
                double reach2 = far2;
                for (uint32_t v = 0; v < mesh.vertex_count; ++v) {
                    const float* p = s.vertices[mesh.vertex_offset + v].p;
                    reach2 = std::max(reach2, (p[0] - c[0]) * (p[0] - c[0]) + (p[1] - c[1]) * (p[1] - c[1]) + (p[2] - c[2]) * (p[2] - c[2]));
                }
                out.mesh_reach[mi] = (float)(std::sqrt(reach2) * 1.0001 + 1e-30);
            }
        }
        out.mesh_root[mi] = (uint32_t)out.blas_pairs.size();
        out.mesh_tri_base[mi] = (uint32_t)(out.blas_tris.size() / 9);
        out.mesh_bounds[mi] = nodes[0];
        flatten(nodes, out.blas_pairs);
        for (uint32_t t = 0; t < ntri; ++t) {
            // stored as its three vertices
            const float* p[3];
            for (int k = 0; k < 3; ++k) p[k] = s.vertices[s.indices[mesh.index_offset + 3 * items[t].index + k] + mesh.vertex_offset].p;
            for (int v = 0; v < 3; ++v)
                for (int c = 0; c < 3; ++c) out.blas_tris.push_back(p[v][c]);
        }
    }
}

void rdr_build_tlas(const rdr_scene_desc& s, const float* transforms, HostBvh& io) {
    std::vector<Item> items(s.draw_count);
    io.instances.resize(s.draw_count);
    io.instance_ball.assign(4 * (size_t)s.draw_count, 0.0f);
    io.instance_reach.assign(s.draw_count, 0.0f);
    for (uint32_t di = 0; di < s.draw_count; ++di) {
        const rdr_mesh_draw& d = s.mesh_draws[di];
        const float* m = transforms + (size_t)d.model_index * 16;
        GpuInstance& inst = io.instances[di];
        affine_inverse(m, inst.inv);
        inst.blas_root = io.mesh_root[d.mesh_index];
        inst.tri_base = io.mesh_tri_base[d.mesh_index];
        inst.outer_radius = io.mesh_ball[5 * (size_t)d.mesh_index + 4];
        io.instance_reach[di] = io.mesh_reach[d.mesh_index];
        inst.convex_cos = 0.0f;
        for (int k = 0; k < 4; ++k) io.instance_ball[4 * (size_t)di + k] = io.mesh_ball[5 * (size_t)d.mesh_index + k];
        const BvhNode& root = io.mesh_bounds[d.mesh_index];
        Item& it = items[di];
        it.box.reset();
        it.index = di;
        if (root.lo[0] <= root.hi[0]) {
            for (int corner = 0; corner < 8; ++corner) {
                const float x = (corner & 1) ? root.hi[0] : root.lo[0];
                const float y = (corner & 2) ? root.hi[1] : root.lo[1];
                const float z = (corner & 4) ? root.hi[2] : root.lo[2];
                const float w[3] = {m[0] * x + m[4] * y + m[8] * z + m[12], m[1] * x + m[5] * y + m[9] * z + m[13],
                                    m[2] * x + m[6] * y + m[10] * z + m[14]};
                it.box.grow(w);
            }
            for (int k = 0; k < 3; ++k) {
                const float pad = 1e-4f * (it.box.hi[k] - it.box.lo[k]) + 1e-6f * (1.0f + std::fabs(it.box.hi[k]) + std::fabs(it.box.lo[k]));
                it.box.lo[k] -= pad;
                it.box.hi[k] += pad;
            }
        } else {
            const float far_away[3] = {3e38f, 3e38f, 3e38f};
            it.box.grow(far_away);
        }
        for (int k = 0; k < 3; ++k) it.c[k] = 0.5f * (it.box.lo[k] + it.box.hi[k]);
        if (io.mesh_convex_sign[d.mesh_index] != 0 && root.lo[0] <= root.hi[0]) {
            // A ray that leaves a face at cosine c to its outward normal is c * tmin above the face plane when the
            // triangle tests start to count (tmin = 0.01).  That clearance must exceed what rounding can move: the
            // origin (interpolated world position) and the object-space triangle tests are good to a few units in
            // the last place of the largest world coordinate M involved (64 of them are allowed for), plus the
            // mesh's own deviation from exact convexity, stretched by the model matrix.
            const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[6] * m[9]) - (double)m[4] * ((double)m[1] * m[10] - (double)m[2] * m[9]) +
                               (double)m[8] * ((double)m[1] * m[6] - (double)m[2] * m[5]);
            double scale = 0.0, M = 0.0;
            {
                // how far the matrix can stretch a vector, from above (largest absolute row sum of A^T A; the longest
                // column is a lower bound once the matrix shears)
                double g[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        g[a][b] = std::fabs((double)m[4 * a] * m[4 * b] + (double)m[4 * a + 1] * m[4 * b + 1] + (double)m[4 * a + 2] * m[4 * b + 2]);
                for (int a = 0; a < 3; ++a) scale = std::max(scale, std::sqrt(g[a][0] + g[a][1] + g[a][2]));
            }
            for (int k = 0; k < 3; ++k) M = std::max(M, std::max(std::fabs((double)it.box.lo[k]), std::fabs((double)it.box.hi[k])));
            const double cos_min = std::max(0.01, (16.0 * 1.1920929e-7 * M + 4.0 * io.mesh_protrusion[d.mesh_index] * scale) / 0.01);
            if (det != 0.0 && std::isfinite(det) && cos_min < 0.95)
                inst.convex_cos = (float)(cos_min * (double)io.mesh_convex_sign[d.mesh_index] * (det > 0.0 ? 1.0 : -1.0));
        }
    }
    build(io.tlas_nodes, items);
    io.tlas_pairs.clear();
    flatten(io.tlas_nodes, io.tlas_pairs);
    io.tlas_items.resize(items.size());
    for (size_t i = 0; i < items.size(); ++i) io.tlas_items[i] = items[i].index;
}
