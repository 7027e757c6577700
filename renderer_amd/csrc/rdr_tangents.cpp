// rdr_tangents.cpp — per-vertex tangents for meshes that come without them, host code.
//
// The reference calls the mikktspace crate (0.3.0, pinned in Cargo.lock; not part of the reference tree) for glTF
// primitives without a TANGENT attribute (scenery/src/vertex.rs:47-90, called at scenery/src/scene.rs:207).  This is a
// restatement of that library's published algorithm (Mikkelsen's tangent space, as in Blender / xNormal / glTF
// sample viewers), triangles only, angular threshold 180 degrees as the crate's generate_tangents() uses:
//   1. corners with identical position, normal and texture coordinate are one vertex;
//   2. per triangle: dP/du and dP/dv from the uv parametrisation, normalised and given the sign of the uv area;
//      a triangle whose uv mapping is degenerate may join any group; a triangle with two equal positions takes no part
//      and borrows its corners' results from a good triangle afterwards;
//   3. around every vertex, triangles connected through shared edges and of the same uv orientation form a group;
//   4. a corner's tangent is the sum over its group of the triangles' dP/du projected into the plane of the vertex
//      normal, each weighted by the triangle's angle at the vertex, normalised; w = +1 where u x v has the orientation
//      of the surface, else -1;
//   5. results are written corner by corner, in triangle order, into the corner's vertex record: as in the reference's
//      set_tangent_encoded, the last corner written to a shared record wins.
// Single precision throughout, as the library computes.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "rdr_internal.h"

namespace {

struct V3 {
    float x, y, z;
};
V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float length(V3 a) { return std::sqrt(dot(a, a)); }
bool not_zero(float x) { return std::fabs(x) > FLT_MIN; }
bool not_zero(V3 a) { return not_zero(a.x) || not_zero(a.y) || not_zero(a.z); }
V3 normalized(V3 a) { return (1.0f / length(a)) * a; }
V3 project(V3 v, V3 n) { // v minus its component along n, normalised when it is not zero
    V3 r = v - dot(n, v) * n;
    return not_zero(r) ? normalized(r) : r;
}

constexpr uint32_t kGroupWithAny = 1u, kOrientPreserving = 2u, kDegenerate = 4u;

struct Tri {
    int neighbor[3] = {-1, -1, -1}; // across edge (i, i + 1)
    int group[3] = {-1, -1, -1};
    V3 os{0, 0, 0}, ot{0, 0, 0};
    float mag_s = 0.0f, mag_t = 0.0f;
    uint32_t flags = kGroupWithAny;
};

struct Group {
    uint32_t vertex;
    bool orient;
    std::vector<int> faces;
};

struct Space {
    V3 os{1.0f, 0.0f, 0.0f};
    bool orient = false;
};

struct Builder {
    const rdr_vertex* verts;
    const uint32_t* weld; // vertex id -> welded id
    std::vector<uint32_t> corner; // welded id per corner (3 per triangle)
    std::vector<uint32_t> original; // vertex id per corner
    std::vector<Tri> tris;
    std::vector<Group> groups;

    V3 position(uint32_t corner_index) const {
        const float* p = verts[original[corner_index]].p;
        return {p[0], p[1], p[2]};
    }
    V3 normal(uint32_t corner_index) const {
        const float* n = verts[original[corner_index]].n;
        return {n[0], n[1], n[2]};
    }

    bool assign(int f, int g) { // the library's AssignRecur, with an explicit stack
        std::vector<int> todo{f};
        bool first = true, answer = false;
        while (!todo.empty()) {
            const int t = todo.back();
            todo.pop_back();
            Tri& T = tris[t];
            int i = -1;
            for (int k = 0; k < 3; ++k)
                if (corner[3 * t + k] == groups[g].vertex) { i = k; break; }
            bool ok = false;
            if (i >= 0 && T.group[i] == g) ok = true;
            else if (i >= 0 && T.group[i] < 0) {
                if (T.flags & kGroupWithAny) {
                    // the first group to claim such a triangle gives it its orientation
                    if (T.group[0] < 0 && T.group[1] < 0 && T.group[2] < 0)
                        T.flags = (T.flags & ~kOrientPreserving) | (groups[g].orient ? kOrientPreserving : 0u);
                }
                if (((T.flags & kOrientPreserving) != 0) == groups[g].orient) {
                    groups[g].faces.push_back(t);
                    T.group[i] = g;
                    const int left = T.neighbor[i], right = T.neighbor[i > 0 ? i - 1 : 2];
                    // the library recurses into the left neighbour first, then the right one
                    if (right >= 0) todo.push_back(right);
                    if (left >= 0) todo.push_back(left);
                    ok = true;
                }
            }
            if (first) { answer = ok; first = false; }
        }
        return answer;
    }

    Space evaluate(const std::vector<int>& faces, uint32_t vertex) const { // the library's EvalTspace
        V3 os{0, 0, 0};
        for (int f : faces) {
            const Tri& T = tris[f];
            if (T.flags & kGroupWithAny) continue;
            int i = -1;
            for (int k = 0; k < 3; ++k)
                if (corner[3 * f + k] == vertex) { i = k; break; }
            if (i < 0) continue;
            const V3 n = normal(3 * f + i);
            const V3 v_os = project(T.os, n);
            const uint32_t c2 = 3 * f + (i < 2 ? i + 1 : 0), c1 = 3 * f + i, c0 = 3 * f + (i > 0 ? i - 1 : 2);
            const V3 p0 = position(c0), p1 = position(c1), p2 = position(c2);
            const V3 v1 = project(p0 - p1, n), v2 = project(p2 - p1, n);
            float c = dot(v1, v2);
            c = c > 1.0f ? 1.0f : (c < -1.0f ? -1.0f : c);
            const float angle = (float)std::acos((double)c);
            os = os + angle * v_os;
        }
        Space s;
        s.os = not_zero(os) ? normalized(os) : os;
        return s;
    }
};

} // namespace

extern "C" int rdr_generate_tangents(rdr_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count) {
    RDR_CHECK(vertices && indices, "rdr_generate_tangents: null argument");
    RDR_CHECK(index_count % 3 == 0, "rdr_generate_tangents: index count is not a multiple of three");
    for (uint32_t i = 0; i < index_count; ++i) RDR_CHECK(indices[i] < vertex_count, "rdr_generate_tangents: index out of range");
    const uint32_t n_tris = index_count / 3;

    // 1. weld: identical position, normal, texture coordinate (bit patterns, -0 = +0)
    std::vector<uint32_t> weld(vertex_count);
    {
        struct Key {
            uint32_t w[8];
            bool operator==(const Key& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
        };
        struct Hash {
            size_t operator()(const Key& k) const {
                uint64_t h = 0xCBF29CE484222325ull;
                for (uint32_t x : k.w) h = (h ^ x) * 0x100000001B3ull;
                return (size_t)h;
            }
        };
        std::unordered_map<Key, uint32_t, Hash> ids;
        ids.reserve((size_t)vertex_count * 2);
        for (uint32_t v = 0; v < vertex_count; ++v) {
            const rdr_vertex& r = vertices[v];
            const float f[8] = {r.p[0] + 0.0f, r.p[1] + 0.0f, r.p[2] + 0.0f, r.n[0] + 0.0f, r.n[1] + 0.0f, r.n[2] + 0.0f, r.uv[0] + 0.0f, r.uv[1] + 0.0f};
            Key k;
            std::memcpy(k.w, f, sizeof(f));
            weld[v] = ids.emplace(k, v).first->second;
        }
    }

    Builder b;
    b.verts = vertices;
    b.weld = weld.data();
    // degenerate triangles (two equal positions) go behind the good ones, which keep their order
    std::vector<uint32_t> order;
    order.reserve(n_tris);
    std::vector<uint32_t> degenerate;
    auto same_position = [&](uint32_t a, uint32_t c) { return std::memcmp(vertices[a].p, vertices[c].p, 12) == 0 || (vertices[a].p[0] == vertices[c].p[0] && vertices[a].p[1] == vertices[c].p[1] && vertices[a].p[2] == vertices[c].p[2]); };
    for (uint32_t t = 0; t < n_tris; ++t) {
        const uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
        if (same_position(i0, i1) || same_position(i0, i2) || same_position(i1, i2)) degenerate.push_back(t);
        else order.push_back(t);
    }
    const uint32_t n_good = (uint32_t)order.size();
    b.tris.resize(n_good);
    b.corner.resize((size_t)3 * n_good);
    b.original.resize((size_t)3 * n_good);
    for (uint32_t t = 0; t < n_good; ++t)
        for (int k = 0; k < 3; ++k) {
            b.original[3 * t + k] = indices[3 * order[t] + k];
            b.corner[3 * t + k] = weld[indices[3 * order[t] + k]];
        }

    // 2. first-order derivatives per triangle
    for (uint32_t t = 0; t < n_good; ++t) {
        Tri& T = b.tris[t];
        const V3 v1 = b.position(3 * t), v2 = b.position(3 * t + 1), v3 = b.position(3 * t + 2);
        const float* t1 = vertices[b.original[3 * t]].uv;
        const float* t2 = vertices[b.original[3 * t + 1]].uv;
        const float* t3 = vertices[b.original[3 * t + 2]].uv;
        const float t21x = t2[0] - t1[0], t21y = t2[1] - t1[1], t31x = t3[0] - t1[0], t31y = t3[1] - t1[1];
        const V3 d1 = v2 - v1, d2 = v3 - v1;
        const float signed_area = t21x * t31y - t21y * t31x;
        V3 os = (t31y * d1) - (t21y * d2);
        V3 ot = ((-t31x) * d1) + (t21x * d2);
        if (signed_area > 0.0f) T.flags |= kOrientPreserving;
        if (not_zero(signed_area)) {
            const float abs_area = std::fabs(signed_area), len_os = length(os), len_ot = length(ot);
            const float s = (T.flags & kOrientPreserving) ? 1.0f : -1.0f;
            if (not_zero(len_os)) os = (s / len_os) * os;
            if (not_zero(len_ot)) ot = (s / len_ot) * ot;
            T.mag_s = len_os / abs_area;
            T.mag_t = len_ot / abs_area;
            if (not_zero(T.mag_s) && not_zero(T.mag_t)) T.flags &= ~kGroupWithAny;
        }
        T.os = os;
        T.ot = ot;
    }

    // neighbours across shared edges (the same two welded vertices): edges sorted, consecutive equal ones paired
    {
        struct Edge {
            uint32_t a, b;
            int f;
            int e;
        };
        std::vector<Edge> edges;
        edges.reserve((size_t)3 * n_good);
        for (uint32_t t = 0; t < n_good; ++t)
            for (int k = 0; k < 3; ++k) {
                const uint32_t i0 = b.corner[3 * t + k], i1 = b.corner[3 * t + (k < 2 ? k + 1 : 0)];
                edges.push_back({std::min(i0, i1), std::max(i0, i1), (int)t, k});
            }
        std::stable_sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.f < y.f); });
        for (size_t i = 0; i < edges.size(); ++i) {
            const Edge& e = edges[i];
            if (b.tris[e.f].neighbor[e.e] >= 0) continue;
            for (size_t j = i + 1; j < edges.size() && edges[j].a == e.a && edges[j].b == e.b; ++j) {
                const Edge& o = edges[j];
                if (o.f != e.f && b.tris[o.f].neighbor[o.e] < 0) {
                    b.tris[e.f].neighbor[e.e] = o.f;
                    b.tris[o.f].neighbor[o.e] = e.f;
                    break;
                }
            }
        }
    }

    // 3. groups around every vertex
    for (uint32_t t = 0; t < n_good; ++t)
        for (int i = 0; i < 3; ++i) {
            Tri& T = b.tris[t];
            if ((T.flags & kGroupWithAny) || T.group[i] >= 0) continue;
            const int g = (int)b.groups.size();
            b.groups.push_back({b.corner[3 * t + i], (T.flags & kOrientPreserving) != 0, {}});
            b.groups[g].faces.push_back((int)t);
            T.group[i] = g;
            const int left = T.neighbor[i], right = T.neighbor[i > 0 ? i - 1 : 2];
            if (left >= 0) b.assign(left, g);
            if (right >= 0) b.assign(right, g);
        }

    // 4. one tangent per (group, sub-group): with the 180 degree threshold two triangles of a group fall into
    // different sub-groups only when their projected derivatives point exactly opposite ways
    std::vector<Space> corner_space((size_t)3 * n_good);
    for (const Group& g : b.groups) {
        std::vector<std::vector<int>> sub_members;
        std::vector<Space> sub_space;
        for (int f : g.faces) {
            int index = -1;
            for (int k = 0; k < 3; ++k)
                if (b.tris[f].group[k] == (int)(&g - b.groups.data())) { index = k; break; }
            if (index < 0) continue;
            const V3 n = b.normal(3 * f + index);
            const V3 os = project(b.tris[f].os, n), ot = project(b.tris[f].ot, n);
            std::vector<int> members;
            for (int t : g.faces) {
                const V3 os2 = project(b.tris[t].os, n), ot2 = project(b.tris[t].ot, n);
                const bool any = ((b.tris[f].flags | b.tris[t].flags) & kGroupWithAny) != 0;
                if (any || t == f || (dot(os, os2) > -1.0f && dot(ot, ot2) > -1.0f)) members.push_back(t);
            }
            std::sort(members.begin(), members.end());
            size_t l = 0;
            while (l < sub_members.size() && sub_members[l] != members) ++l;
            if (l == sub_members.size()) {
                sub_space.push_back(b.evaluate(members, g.vertex));
                sub_members.push_back(members);
            }
            Space s = sub_space[l];
            s.orient = g.orient;
            corner_space[3 * f + index] = s;
        }
    }

    // 5. write corner by corner in triangle order; corners of degenerate triangles borrow from a good triangle's
    // corner at the same welded vertex (or keep the default)
    std::vector<int> good_corner_of(vertex_count, -1);
    for (uint32_t c = 0; c < 3 * n_good; ++c)
        if (good_corner_of[b.corner[c]] < 0) good_corner_of[b.corner[c]] = (int)c;
    std::vector<uint32_t> position_in_order(n_tris, 0xFFFFFFFFu);
    for (uint32_t t = 0; t < n_good; ++t) position_in_order[order[t]] = t;
    for (uint32_t t = 0; t < n_tris; ++t)
        for (int k = 0; k < 3; ++k) {
            Space s;
            if (position_in_order[t] != 0xFFFFFFFFu) {
                s = corner_space[3 * position_in_order[t] + k];
            } else {
                const int c = good_corner_of[weld[indices[3 * t + k]]];
                if (c >= 0) s = corner_space[c];
            }
            float* out = vertices[indices[3 * t + k]].t;
            out[0] = s.os.x; out[1] = s.os.y; out[2] = s.os.z;
            out[3] = s.orient ? 1.0f : -1.0f;
        }
    return 0;
}
