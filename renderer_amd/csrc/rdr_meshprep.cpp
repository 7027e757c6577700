// rdr_meshprep.cpp — host-side mesh preparation: what Geometry::append_mesh does with the
// meshoptimizer library before anything is uploaded (scenery/src/mesh.rs:197-374):
//   vertex de-duplication, vertex-cache and vertex-fetch ordering, meshlets of at most
//   64 vertices / 124 triangles with their bounding sphere and normal cone, and a chain of
//   simplified levels of detail (index target x0.65 per level, error limit 0.1 of the extent).
// meshoptimizer itself is not part of this tree; the algorithms below are this library's own
// (linear-speed vertex cache ordering, greedy adjacency-driven meshlet growth, quadric-error
// edge collapse onto existing vertices), so meshlet contents and LOD errors follow the same
// rules and limits as the reference's but are not bit-identical to it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <unordered_map>

#include "rdr_internal.h"

namespace {

constexpr uint32_t kMaxVertices = 64, kMaxTriangles = 124; // scenery/src/mesh.rs:213-214
constexpr float kConeWeight = 0.25f;                       // scenery/src/mesh.rs:215
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct P3 {
    double x, y, z;
};
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator*(P3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline P3 pos_of(const rdr_vertex& v) { return {v.p[0], v.p[1], v.p[2]}; }

uint64_t hash_bytes(const void* p, size_t n) { // FNV-1a
    const uint8_t* b = static_cast<const uint8_t*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// ---- 1. de-duplication: identical vertices collapse, numbered by first use ------------------
void dedupe(const rdr_vertex* v, uint32_t nv, std::vector<uint32_t>& idx, std::vector<rdr_vertex>& out) {
    size_t cap = 16;
    while (cap < (size_t)nv * 2) cap <<= 1;
    std::vector<uint32_t> table(cap, kNone), remap(nv, kNone);
    out.clear();
    for (uint32_t& i : idx) {
        if (remap[i] == kNone) {
            size_t h = hash_bytes(&v[i], sizeof(rdr_vertex)) & (cap - 1);
            while (table[h] != kNone && std::memcmp(&out[table[h]], &v[i], sizeof(rdr_vertex)) != 0) h = (h + 1) & (cap - 1);
            if (table[h] == kNone) {
                table[h] = (uint32_t)out.size();
                out.push_back(v[i]);
            }
            remap[i] = table[h];
        }
        i = remap[i];
    }
}

// ---- 2. vertex cache ordering (linear-speed, score driven) ----------------------------------
// A vertex scores by its position in a simulated LRU cache and by how few triangles still need
// it; the next triangle is the best-scoring one among those touching the cache.
void optimize_vertex_cache(std::vector<uint32_t>& idx, uint32_t nv) {
    const uint32_t nt = (uint32_t)idx.size() / 3;
    if (nt == 0) return;
    constexpr int kCache = 16;
    std::vector<uint32_t> live(nv, 0), first(nv + 1, 0);
    for (uint32_t i : idx) live[i]++;
    for (uint32_t v = 0; v < nv; ++v) first[v + 1] = first[v] + live[v];
    std::vector<uint32_t> fill(first.begin(), first.end() - 1), adj(idx.size());
    for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) adj[fill[idx[t * 3 + k]]++] = t;
    std::vector<int> cache_pos(nv, -1);
    std::vector<float> vscore(nv), tscore(nt, 0.0f);
    auto score_of = [&](uint32_t v) {
        if (live[v] == 0) return -1.0f;
        float s = 0.0f;
        const int p = cache_pos[v];
        if (p >= 0) s = p < 3 ? 0.75f : std::pow(1.0f - (float)(p - 3) / (float)(kCache - 3), 1.5f);
        return s + 2.0f * std::pow((float)live[v], -0.5f);
    };
    for (uint32_t v = 0; v < nv; ++v) vscore[v] = score_of(v);
    for (uint32_t t = 0; t < nt; ++t) tscore[t] = vscore[idx[t * 3]] + vscore[idx[t * 3 + 1]] + vscore[idx[t * 3 + 2]];
    std::vector<uint8_t> done(nt, 0);
    std::vector<uint32_t> out;
    out.reserve(idx.size());
    uint32_t cache[kCache + 3], cache_n = 0, cursor = 0;
    int best = -1;
    for (uint32_t emitted = 0; emitted < nt; ++emitted) {
        if (best < 0) { // nothing near the cache: take the next triangle not yet emitted
            while (done[cursor]) cursor++;
            best = (int)cursor;
        }
        const uint32_t t = (uint32_t)best;
        done[t] = 1;
        uint32_t tri[3] = {idx[t * 3], idx[t * 3 + 1], idx[t * 3 + 2]};
        for (uint32_t v : tri) {
            out.push_back(v);
            live[v]--;
        }
        // new cache: the triangle's vertices in front, the old ones behind
        uint32_t next[kCache + 3], n = 0;
        for (uint32_t v : tri) next[n++] = v;
        for (uint32_t i = 0; i < cache_n; ++i)
            if (cache[i] != tri[0] && cache[i] != tri[1] && cache[i] != tri[2]) next[n++] = cache[i];
        for (uint32_t i = 0; i < n; ++i) cache_pos[next[i]] = i < (uint32_t)kCache ? (int)i : -1;
        cache_n = std::min<uint32_t>(n, kCache);
        std::memcpy(cache, next, sizeof(uint32_t) * cache_n);
        best = -1;
        float best_score = -1.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t v = next[i];
            const float s = score_of(v), delta = s - vscore[v];
            vscore[v] = s;
            for (uint32_t a = first[v]; a < first[v + 1]; ++a) {
                const uint32_t u = adj[a];
                if (done[u]) continue;
                tscore[u] += delta;
                if (tscore[u] > best_score) {
                    best_score = tscore[u];
                    best = (int)u;
                }
            }
        }
    }
    idx.swap(out);
}

// ---- 2b. vertex cache ordering for a FIFO cache (the reference's fast mode) --------------------
// Linear-time fanning: emit every remaining triangle around a vertex, then continue with the vertex
// of that fan that will still be in a FIFO cache of `cache_size` entries after its own triangles are
// emitted (the one that entered the cache longest ago among those); on a dead end fall back to the
// most recently touched vertex that still has triangles, then to the next untouched one.
void optimize_vertex_cache_fifo(std::vector<uint32_t>& idx, uint32_t nv, uint32_t cache_size) {
    const uint32_t nt = (uint32_t)idx.size() / 3;
    if (nt == 0) return;
    std::vector<uint32_t> live(nv, 0), first(nv + 1, 0);
    for (uint32_t i : idx) live[i]++;
    for (uint32_t v = 0; v < nv; ++v) first[v + 1] = first[v] + live[v];
    std::vector<uint32_t> fill(first.begin(), first.end() - 1), adj(idx.size());
    for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) adj[fill[idx[t * 3 + k]]++] = t;
    std::vector<uint32_t> stamp(nv, 0), dead_end, out, ring;
    std::vector<uint8_t> done(nt, 0);
    out.reserve(idx.size());
    uint32_t now = cache_size + 1, cursor = 0;
    int fan = (int)idx[0];
    while (fan >= 0) {
        ring.clear();
        for (uint32_t a = first[fan]; a < first[fan + 1]; ++a) {
            const uint32_t t = adj[a];
            if (done[t]) continue;
            done[t] = 1;
            for (int k = 0; k < 3; ++k) {
                const uint32_t v = idx[t * 3 + k];
                out.push_back(v);
                dead_end.push_back(v);
                ring.push_back(v);
                live[v]--;
                if (now - stamp[v] > cache_size) stamp[v] = now++;
            }
        }
        int next = -1;
        uint32_t best = 0;
        for (uint32_t v : ring) {
            if (live[v] == 0) continue;
            uint32_t age = 0; // position in the cache once v's own triangles have gone through it
            if (now - stamp[v] + 2 * live[v] <= cache_size) age = now - stamp[v];
            if (next < 0 || age > best) { best = age; next = (int)v; }
        }
        if (next < 0) {
            while (!dead_end.empty() && next < 0) {
                const uint32_t v = dead_end.back();
                dead_end.pop_back();
                if (live[v] > 0) next = (int)v;
            }
            while (next < 0 && cursor < nv) {
                if (live[cursor] > 0) next = (int)cursor;
                else cursor++;
            }
        }
        fan = next;
    }
    idx.swap(out);
}

// ---- 3. vertex fetch ordering: vertices in order of first use --------------------------------
void optimize_vertex_fetch(std::vector<uint32_t>& idx, std::vector<rdr_vertex>& verts) {
    std::vector<uint32_t> remap(verts.size(), kNone);
    std::vector<rdr_vertex> out;
    out.reserve(verts.size());
    for (uint32_t& i : idx) {
        if (remap[i] == kNone) {
            remap[i] = (uint32_t)out.size();
            out.push_back(verts[i]);
        }
        i = remap[i];
    }
    verts.swap(out);
}

// ---- 4. meshlets -----------------------------------------------------------------------------
struct MeshletSet {
    std::vector<uint32_t> vertices;  // mesh-local vertex ids
    std::vector<uint8_t> triangles;  // 3 bytes per triangle, each meshlet padded to 4
    std::vector<rdr_meshlet> meshlets;
};

// Bounding sphere (two-pass: extremal pair, then grow) and normal cone, quantised so that the
// signed-byte cone still contains every triangle normal (the fields of scenery/src/mesh.rs:17-28).
void meshlet_bounds(const std::vector<rdr_vertex>& verts, const uint32_t* mv, uint32_t nv, const uint8_t* tris, uint32_t nt, rdr_meshlet& m) {
    P3 a = pos_of(verts[mv[0]]), b = a;
    double far_d = -1.0;
    for (uint32_t i = 0; i < nv; ++i) {
        const P3 d = pos_of(verts[mv[i]]) - pos_of(verts[mv[0]]);
        if (dot(d, d) > far_d) { far_d = dot(d, d); a = pos_of(verts[mv[i]]); }
    }
    far_d = -1.0;
    for (uint32_t i = 0; i < nv; ++i) {
        const P3 d = pos_of(verts[mv[i]]) - a;
        if (dot(d, d) > far_d) { far_d = dot(d, d); b = pos_of(verts[mv[i]]); }
    }
    P3 c = (a + b) * 0.5;
    double r = std::sqrt(dot(a - b, a - b)) * 0.5;
    for (uint32_t i = 0; i < nv; ++i) {
        const P3 q = pos_of(verts[mv[i]]);
        const double d = std::sqrt(dot(q - c, q - c));
        if (d > r) {
            const double nr = (r + d) * 0.5;
            c = c + (q - c) * ((nr - r) / d);
            r = nr;
        }
    }
    m.center[0] = (float)c.x; m.center[1] = (float)c.y; m.center[2] = (float)c.z;
    m.radius = (float)(r * (1.0 + 1e-6) + 1e-12);
    // the float centre moved by up to an ulp: make sure the float sphere still holds every vertex
    for (uint32_t i = 0; i < nv; ++i) {
        const P3 q = pos_of(verts[mv[i]]) - P3{m.center[0], m.center[1], m.center[2]};
        const float d = (float)std::sqrt(dot(q, q));
        if (d > m.radius) m.radius = std::nextafter(d, INFINITY);
    }
    m.cone_axis[0] = m.cone_axis[1] = m.cone_axis[2] = 0;
    m.cone_cutoff = 127; // "never cull"
    P3 axis{0, 0, 0};
    std::vector<P3> normals;
    for (uint32_t t = 0; t < nt; ++t) {
        const P3 p0 = pos_of(verts[mv[tris[t * 3]]]), p1 = pos_of(verts[mv[tris[t * 3 + 1]]]), p2 = pos_of(verts[mv[tris[t * 3 + 2]]]);
        P3 n = cross(p1 - p0, p2 - p0);
        const double l = std::sqrt(dot(n, n));
        if (l > 0.0) {
            n = n * (1.0 / l);
            normals.push_back(n);
            axis = axis + n;
        }
    }
    const double la = std::sqrt(dot(axis, axis));
    if (normals.empty() || la < 1e-12) return;
    axis = axis * (1.0 / la);
    double mindp = 1.0;
    for (const P3& n : normals) mindp = std::min(mindp, dot(n, axis));
    if (mindp <= 0.1) return;
    const double cutoff = std::sqrt(std::max(0.0, 1.0 - mindp * mindp));
    const double q[3] = {std::round(axis.x * 127.0), std::round(axis.y * 127.0), std::round(axis.z * 127.0)};
    const double err = std::fabs(q[0] / 127.0 - axis.x) + std::fabs(q[1] / 127.0 - axis.y) + std::fabs(q[2] / 127.0 - axis.z);
    for (int k = 0; k < 3; ++k) m.cone_axis[k] = (int8_t)std::max(-127.0, std::min(127.0, q[k]));
    m.cone_cutoff = (int8_t)std::min(127, (int)(127.0 * cutoff + 127.0 * err + 1.0));
}

// Greedy growth: a meshlet starts at the first unused triangle (index order, which the cache
// ordering made spatially coherent) and keeps taking the unused neighbour that needs the fewest
// new vertices, then lies closest to the meshlet and agrees most with its average normal (the
// cone weight trades the two), until a limit would be exceeded or no neighbour is left.
void build_meshlets(const std::vector<uint32_t>& idx, const std::vector<rdr_vertex>& verts, MeshletSet& out) {
    const uint32_t nt = (uint32_t)idx.size() / 3, nv = (uint32_t)verts.size();
    std::vector<uint32_t> first(nv + 1, 0);
    for (uint32_t i : idx) first[i + 1]++;
    for (uint32_t v = 0; v < nv; ++v) first[v + 1] += first[v];
    std::vector<uint32_t> fill(first.begin(), first.end() - 1), adj(idx.size());
    for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) adj[fill[idx[t * 3 + k]]++] = t;
    std::vector<P3> centroid(nt), normal(nt);
    double area_sum = 0.0;
    for (uint32_t t = 0; t < nt; ++t) {
        const P3 p0 = pos_of(verts[idx[t * 3]]), p1 = pos_of(verts[idx[t * 3 + 1]]), p2 = pos_of(verts[idx[t * 3 + 2]]);
        centroid[t] = (p0 + p1 + p2) * (1.0 / 3.0);
        P3 n = cross(p1 - p0, p2 - p0);
        const double l = std::sqrt(dot(n, n));
        area_sum += l * 0.5;
        normal[t] = l > 0.0 ? n * (1.0 / l) : P3{0, 0, 0};
    }
    // radius a full meshlet of average triangles would have: the unit of the distance score
    const double expected_r2 = std::max(1e-30, area_sum / std::max(1u, nt) * kMaxTriangles * 0.5);
    std::vector<uint8_t> used(nt, 0), slot(nv, 0xFF);
    std::vector<uint32_t> live(nv); // unused triangles around a vertex
    for (uint32_t v = 0; v < nv; ++v) live[v] = first[v + 1] - first[v];
    uint32_t cursor = 0;
    std::vector<uint32_t> mv;
    std::vector<uint8_t> mt;
    P3 csum{0, 0, 0}, nsum{0, 0, 0};
    auto flush = [&]() {
        if (mt.empty()) return;
        rdr_meshlet m{};
        m.vertex_offset = (uint32_t)out.vertices.size();
        m.triangle_offset = (uint32_t)out.triangles.size();
        m.vertex_count = (uint8_t)mv.size();
        m.triangle_count = (uint8_t)(mt.size() / 3);
        meshlet_bounds(verts, mv.data(), (uint32_t)mv.size(), mt.data(), (uint32_t)mt.size() / 3, m);
        out.vertices.insert(out.vertices.end(), mv.begin(), mv.end());
        out.triangles.insert(out.triangles.end(), mt.begin(), mt.end());
        while (out.triangles.size() % 4) out.triangles.push_back(0);
        out.meshlets.push_back(m);
        for (uint32_t v : mv) slot[v] = 0xFF;
        mv.clear();
        mt.clear();
        csum = nsum = P3{0, 0, 0};
    };
    auto take = [&](uint32_t t) {
        used[t] = 1;
        for (int k = 0; k < 3; ++k) {
            const uint32_t v = idx[t * 3 + k];
            if (slot[v] == 0xFF) {
                slot[v] = (uint8_t)mv.size();
                mv.push_back(v);
            }
            live[v]--;
            mt.push_back(slot[v]);
        }
        csum = csum + centroid[t];
        nsum = nsum + normal[t];
    };
    for (uint32_t taken = 0; taken < nt;) {
        uint32_t best = kNone, best_extra = 4;
        double best_score = 0.0;
        if (!mt.empty()) {
            const double inv_n = 1.0 / (double)(mt.size() / 3);
            const P3 c = csum * inv_n;
            const double nl = std::sqrt(dot(nsum, nsum));
            const P3 axis = nl > 0.0 ? nsum * (1.0 / nl) : P3{0, 0, 0};
            for (uint32_t v : mv) {
                if (live[v] == 0) continue;
                for (uint32_t a = first[v]; a < first[v + 1]; ++a) {
                    const uint32_t t = adj[a];
                    if (used[t]) continue;
                    uint32_t extra = 0;
                    for (int k = 0; k < 3; ++k) extra += slot[idx[t * 3 + k]] == 0xFF;
                    // a vertex whose last unused triangle this is costs nothing to finish off
                    for (int k = 0; k < 3; ++k)
                        if (slot[idx[t * 3 + k]] != 0xFF && live[idx[t * 3 + k]] == 1 && extra > 0) { extra--; break; }
                    const P3 d = centroid[t] - c;
                    const double spatial = dot(d, d) / expected_r2, cone = 1.0 - dot(normal[t], axis);
                    const double score = spatial * (1.0 - kConeWeight) + cone * kConeWeight;
                    if (extra < best_extra || (extra == best_extra && score < best_score)) {
                        best = t; best_extra = extra; best_score = score;
                    }
                }
            }
        }
        if (best != kNone) {
            uint32_t extra = 0;
            for (int k = 0; k < 3; ++k) extra += slot[idx[best * 3 + k]] == 0xFF;
            // full: close the meshlet; the neighbour seeds the next one, which keeps growth contiguous
            if (mv.size() + extra > kMaxVertices || mt.size() / 3 + 1 > kMaxTriangles) flush();
        }
        if (best == kNone) {
            if (!mt.empty()) flush(); // no neighbour left: the meshlet is closed
            while (used[cursor]) cursor++;
            best = cursor;
        }
        take(best);
        taken++;
    }
    flush();
}

// ---- 5. simplification -------------------------------------------------------------------------
// Quadric-error edge collapse onto existing vertices (no new positions, so every level shares the
// mesh's vertex buffer).  Works in coordinates divided by the mesh extent, so errors are relative.
struct Quadric {
    double a00 = 0, a11 = 0, a22 = 0, a10 = 0, a20 = 0, a21 = 0, b0 = 0, b1 = 0, b2 = 0, c = 0, w = 0;
    void add(const Quadric& q) {
        a00 += q.a00; a11 += q.a11; a22 += q.a22; a10 += q.a10; a20 += q.a20; a21 += q.a21;
        b0 += q.b0; b1 += q.b1; b2 += q.b2; c += q.c; w += q.w;
    }
    static Quadric plane(P3 n, double d, double weight) {
        Quadric q;
        q.a00 = n.x * n.x * weight; q.a11 = n.y * n.y * weight; q.a22 = n.z * n.z * weight;
        q.a10 = n.x * n.y * weight; q.a20 = n.x * n.z * weight; q.a21 = n.y * n.z * weight;
        q.b0 = n.x * d * weight; q.b1 = n.y * d * weight; q.b2 = n.z * d * weight;
        q.c = d * d * weight; q.w = weight;
        return q;
    }
    double error(P3 p) const { // weighted mean squared distance to the accumulated planes
        const double r = p.x * (a00 * p.x + 2.0 * (a10 * p.y + a20 * p.z + b0)) + p.y * (a11 * p.y + 2.0 * (a21 * p.z + b1)) +
                         p.z * (a22 * p.z + 2.0 * b2) + c;
        return w > 0.0 ? std::fabs(r) / w : 0.0;
    }
};

float simplify_scale(const std::vector<rdr_vertex>& verts) { // largest extent of the bounding box
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const rdr_vertex& v : verts)
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v.p[k]); hi[k] = std::max(hi[k], v.p[k]); }
    return verts.empty() ? 0.0f : std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
}

enum : uint8_t { kManifold = 0, kBorder = 1, kLocked = 2 };

inline uint64_t edge_key(uint32_t a, uint32_t b) { return ((uint64_t)std::min(a, b) << 32) | std::max(a, b); }

// Returns the simplified index list (a subset of `idx`'s vertices) and the relative error reached.
std::vector<uint32_t> simplify(const std::vector<uint32_t>& idx, const std::vector<rdr_vertex>& verts, size_t target_index_count,
                               double error_limit, float* result_error) {
    const uint32_t nv = (uint32_t)verts.size();
    const double scale = simplify_scale(verts), inv_scale = scale > 0.0 ? 1.0 / scale : 1.0;
    std::vector<P3> pos(nv);
    for (uint32_t v = 0; v < nv; ++v) pos[v] = pos_of(verts[v]) * inv_scale;
    // vertices that share a position but differ in attributes (uv seams, hard edges) form a wedge
    // group; the group's members stay where they are
    std::vector<uint32_t> canon(nv);
    {
        std::unordered_map<uint64_t, std::vector<uint32_t>> buckets;
        buckets.reserve(nv * 2);
        for (uint32_t v = 0; v < nv; ++v) {
            std::vector<uint32_t>& b = buckets[hash_bytes(verts[v].p, 12)];
            canon[v] = v;
            for (uint32_t u : b)
                if (std::memcmp(verts[u].p, verts[v].p, 12) == 0) { canon[v] = u; break; }
            if (canon[v] == v) b.push_back(v);
        }
    }
    std::vector<uint8_t> seam(nv, 0);
    for (uint32_t v = 0; v < nv; ++v)
        if (canon[v] != v) seam[v] = seam[canon[v]] = 1;

    std::vector<uint32_t> tri(idx);
    std::vector<Quadric> Q(nv);
    double max_error = 0.0;
    const double limit2 = error_limit * error_limit;
    bool first_pass = true;
    std::vector<uint8_t> kind(nv);
    for (int pass = 0; pass < 64 && tri.size() > target_index_count; ++pass) {
        const uint32_t nt = (uint32_t)tri.size() / 3;
        // topology of the current mesh over positions: an edge used once is a border
        std::unordered_map<uint64_t, uint32_t> edges;
        edges.reserve(tri.size() * 2);
        for (uint32_t t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) edges[edge_key(canon[tri[t * 3 + k]], canon[tri[t * 3 + (k + 1) % 3]])]++;
        std::vector<uint8_t> border_edges(nv, 0), odd(nv, 0);
        for (const auto& e : edges) {
            const uint32_t a = (uint32_t)(e.first >> 32), b = (uint32_t)e.first;
            if (e.second == 1) { border_edges[a] = std::min(255, border_edges[a] + 1); border_edges[b] = std::min(255, border_edges[b] + 1); }
            else if (e.second != 2) odd[a] = odd[b] = 1;
        }
        for (uint32_t v = 0; v < nv; ++v) {
            const uint32_t c = canon[v];
            kind[v] = (seam[v] || odd[c]) ? kLocked : border_edges[c] == 0 ? kManifold : border_edges[c] == 2 ? kBorder : kLocked;
        }
        if (first_pass) { // quadrics of the original surface; collapses add the source's quadric to the target's
            first_pass = false;
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t i0 = tri[t * 3], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
                P3 n = cross(pos[i1] - pos[i0], pos[i2] - pos[i0]);
                const double l = std::sqrt(dot(n, n));
                if (l <= 0.0) continue;
                n = n * (1.0 / l);
                const Quadric q = Quadric::plane(n, -dot(n, pos[i0]), l * 0.5);
                Q[i0].add(q); Q[i1].add(q); Q[i2].add(q);
                for (int k = 0; k < 3; ++k) { // a border edge also resists moving sideways
                    const uint32_t a = tri[t * 3 + k], b = tri[t * 3 + (k + 1) % 3];
                    if (edges[edge_key(canon[a], canon[b])] != 1) continue;
                    const P3 e = pos[b] - pos[a];
                    P3 en = cross(e, n);
                    const double el = std::sqrt(dot(en, en));
                    if (el <= 0.0) continue;
                    en = en * (1.0 / el);
                    const Quadric bq = Quadric::plane(en, -dot(en, pos[a]), dot(e, e) * 10.0);
                    Q[a].add(bq); Q[b].add(bq);
                }
            }
        }
        // adjacency for the flip test
        std::vector<uint32_t> first(nv + 1, 0);
        for (uint32_t i : tri) first[i + 1]++;
        for (uint32_t v = 0; v < nv; ++v) first[v + 1] += first[v];
        std::vector<uint32_t> fill(first.begin(), first.end() - 1), adj(tri.size());
        for (uint32_t t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) adj[fill[tri[t * 3 + k]]++] = t;
        // candidate collapses a -> b along the edges of the mesh
        struct Collapse { uint32_t a, b; double cost; };
        std::vector<Collapse> cand;
        cand.reserve(edges.size());
        auto consider = [&](uint32_t a, uint32_t b, bool is_border_edge) {
            if (kind[a] == kLocked) return;
            if (kind[a] == kBorder && !is_border_edge) return; // a border vertex only slides along its border
            Quadric q = Q[a];
            q.add(Q[b]);
            cand.push_back({a, b, q.error(pos[b])});
        };
        for (uint32_t t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint32_t a = tri[t * 3 + k], b = tri[t * 3 + (k + 1) % 3];
                if (a == b) continue;
                const uint32_t uses = edges[edge_key(canon[a], canon[b])];
                if (a < b || uses == 1) { // each interior edge once (from its a < b side), border edges from their only side
                    consider(a, b, uses == 1);
                    consider(b, a, uses == 1);
                }
            }
        std::sort(cand.begin(), cand.end(), [](const Collapse& x, const Collapse& y) {
            return x.cost != y.cost ? x.cost < y.cost : (x.a != y.a ? x.a < y.a : x.b < y.b);
        });
        std::vector<uint32_t> moved(nv);
        for (uint32_t v = 0; v < nv; ++v) moved[v] = v;
        std::vector<uint8_t> touched(nv, 0);
        size_t tri_count = nt, collapses = 0;
        // keep some headroom per pass so that one pass does not overshoot the target by much
        for (const Collapse& c : cand) {
            if (tri_count * 3 <= target_index_count) break;
            if (c.cost > limit2) break;
            if (touched[c.a] || touched[c.b]) continue;
            // flip test on the triangles around a that survive the collapse
            bool flips = false;
            size_t removed = 0;
            for (uint32_t k = first[c.a]; k < first[c.a + 1] && !flips; ++k) {
                const uint32_t t = adj[k];
                uint32_t v[3] = {tri[t * 3], tri[t * 3 + 1], tri[t * 3 + 2]};
                if (v[0] == c.b || v[1] == c.b || v[2] == c.b) { removed++; continue; }
                const P3 n0 = cross(pos[v[1]] - pos[v[0]], pos[v[2]] - pos[v[0]]);
                for (uint32_t& x : v) if (x == c.a) x = c.b;
                const P3 n1 = cross(pos[v[1]] - pos[v[0]], pos[v[2]] - pos[v[0]]);
                flips = dot(n0, n1) <= 0.25 * std::sqrt(dot(n0, n0) * dot(n1, n1));
            }
            if (flips || removed == 0) continue;
            moved[c.a] = c.b;
            touched[c.a] = touched[c.b] = 1;
            // the ring around a changes shape: leave it alone for the rest of this pass
            for (uint32_t k = first[c.a]; k < first[c.a + 1]; ++k)
                for (int j = 0; j < 3; ++j) touched[tri[adj[k] * 3 + j]] = 1;
            Q[c.b].add(Q[c.a]);
            max_error = std::max(max_error, c.cost);
            tri_count -= removed;
            collapses++;
        }
        if (collapses == 0) break;
        std::vector<uint32_t> next;
        next.reserve(tri.size());
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t a = moved[tri[t * 3]], b = moved[tri[t * 3 + 1]], c = moved[tri[t * 3 + 2]];
            if (a == b || b == c || a == c) continue;
            next.push_back(a); next.push_back(b); next.push_back(c);
        }
        tri.swap(next);
    }
    *result_error = (float)std::sqrt(max_error);
    return tri;
}

} // namespace

struct rdr_mesh_builder {
    std::vector<rdr_vertex> vertices;
    std::vector<uint32_t> indices;
    MeshletSet set;
    rdr_mesh mesh{};
};

extern "C" {

int rdr_mesh_build(const rdr_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                   uint32_t max_lods, uint32_t fast_mode, rdr_mesh_builder** out) {
    RDR_CHECK(out, "rdr_mesh_build: out is null");
    *out = nullptr;
    RDR_CHECK(vertices && indices && vertex_count && index_count, "rdr_mesh_build: empty mesh");
    RDR_CHECK(index_count % 3 == 0, "rdr_mesh_build: index count is not a multiple of 3");
    RDR_CHECK(max_lods >= 1 && max_lods <= 8, "rdr_mesh_build: max_lods must be 1..8");
    for (uint32_t i = 0; i < index_count; ++i) RDR_CHECK(indices[i] < vertex_count, "rdr_mesh_build: index out of range");
    std::unique_ptr<rdr_mesh_builder> b(new rdr_mesh_builder());
    b->indices.assign(indices, indices + index_count);
    dedupe(vertices, vertex_count, b->indices, b->vertices);
    // scenery/src/mesh.rs:204-210: one ordering pass, then a second one whose kind the mode picks
    optimize_vertex_cache(b->indices, (uint32_t)b->vertices.size());
    if (fast_mode) optimize_vertex_cache_fifo(b->indices, (uint32_t)b->vertices.size(), 16);
    else optimize_vertex_cache(b->indices, (uint32_t)b->vertices.size());
    optimize_vertex_fetch(b->indices, b->vertices);

    rdr_mesh& mesh = b->mesh;
    { // centre = mean of the vertices, radius = farthest vertex (scenery/src/mesh.rs:234-243)
        float c[3] = {0, 0, 0};
        for (const rdr_vertex& v : b->vertices) { c[0] += v.p[0]; c[1] += v.p[1]; c[2] += v.p[2]; }
        const float n = (float)b->vertices.size();
        for (int k = 0; k < 3; ++k) mesh.center[k] = c[k] / n;
        float r = 0.0f;
        for (const rdr_vertex& v : b->vertices) {
            const float dx = v.p[0] - mesh.center[0], dy = v.p[1] - mesh.center[1], dz = v.p[2] - mesh.center[2];
            r = std::max(r, std::sqrt((dx * dx + dy * dy) + dz * dz));
        }
        mesh.radius = r;
    }
    mesh.vertex_offset = 0;
    mesh.vertex_count = (uint32_t)b->vertices.size();
    mesh.index_offset = 0;
    mesh.lod_count = 1;
    build_meshlets(b->indices, b->vertices, b->set);
    mesh.lods[0].meshlet_offset = 0;
    mesh.lods[0].meshlet_count = (uint32_t)b->set.meshlets.size();
    mesh.lods[0].error = 0.0f;

    // LOD chain (scenery/src/mesh.rs:271-342): every level is simplified from level 0 towards
    // 0.65^k of its index count with a relative error limit of 0.1; the chain stops when a level
    // keeps 95 % or more of level 0's indices or comes out empty.
    const float lod_scale = simplify_scale(b->vertices);
    size_t target = (size_t)((double)b->indices.size() * 0.65);
    while (mesh.lod_count < max_lods) {
        float error = 0.0f;
        std::vector<uint32_t> lod = simplify(b->indices, b->vertices, target - target % 3, 1e-1, &error);
        if (lod.size() >= (size_t)((double)b->indices.size() * 0.95) || lod.empty()) break;
        if (fast_mode) optimize_vertex_cache_fifo(lod, (uint32_t)b->vertices.size(), 16); // scenery/src/mesh.rs:305-309
        else optimize_vertex_cache(lod, (uint32_t)b->vertices.size());
        const uint32_t before = (uint32_t)b->set.meshlets.size();
        build_meshlets(lod, b->vertices, b->set);
        rdr_mesh_lod& L = mesh.lods[mesh.lod_count];
        L.meshlet_offset = before;
        L.meshlet_count = (uint32_t)b->set.meshlets.size() - before;
        L.error = std::max(error, mesh.lods[mesh.lod_count - 1].error) * lod_scale;
        mesh.lod_count++;
        target = (size_t)((double)target * 0.65);
    }
    *out = b.release();
    return 0;
}

int rdr_mesh_result(const rdr_mesh_builder* b, rdr_built_mesh* out) {
    RDR_CHECK(b && out, "rdr_mesh_result: null argument");
    out->vertices = b->vertices.data();
    out->indices = b->indices.data();
    out->meshlet_vertices = b->set.vertices.data();
    out->meshlet_triangles = b->set.triangles.data();
    out->meshlets = b->set.meshlets.data();
    out->vertex_count = (uint32_t)b->vertices.size();
    out->index_count = (uint32_t)b->indices.size();
    out->meshlet_vertex_count = (uint32_t)b->set.vertices.size();
    out->meshlet_triangle_bytes = (uint32_t)b->set.triangles.size();
    out->meshlet_count = (uint32_t)b->set.meshlets.size();
    out->mesh = b->mesh;
    return 0;
}

void rdr_mesh_free(rdr_mesh_builder* b) { delete b; }

} // extern "C"
