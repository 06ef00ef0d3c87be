// Support-point mesh: a host-side Delaunay triangulation of integer pixel positions feeding the
// device plane fit and rasteriser.  It takes the place of the Fade_2D object of
// highPerfStereo/highPerfStereoLib.cpp:434-445 (insert) and :457 (getTrianglePointers); Fade2D is a
// closed library, so only the defining property is reproduced: no point lies strictly inside a
// triangle's circumcircle.  Where four or more points are cocircular the diagonal is a free choice and
// may differ from Fade2D's; here it is fixed by a symbolic perturbation (the point that comes first in the
// caller's list is lifted by the largest infinitesimal), so the mesh is a function of the point list alone,
// not of the order of construction: inserting points into an existing mesh and triangulating the whole list
// from scratch give the same triangles.
//
// Construction: points are added one at a time in order of their distance from a central input point; each
// new point lies outside the hull built so far and is joined to every hull edge it sees (found from a table
// of hull vertices by direction), then the new edges are legalised by flips.  Later points (the resampling
// step's, highPerfStereoLib.cpp:141-175) are located by a walk from the last triangle touched and split the
// triangle or edge they fall on, or extend the hull.
// All predicates are exact (64/128-bit integers; coordinates must fit in 24 bits).
#include "svx_common.hpp"
#include <algorithm>
#include <cmath>
#include <new>
#include <numeric>
#include <vector>

namespace {

typedef long long i64;
typedef __int128 i128;

struct Pt { i64 x, y; int id; i64 d2; };    // d2: squared distance from the centre of the radial order (set by build)

inline i64 orient(const Pt &a, const Pt &b, const Pt &c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

// > 0 when d lies strictly inside the circle through a, b, c (a, b, c counter-clockwise).
inline int in_circle(const Pt &a, const Pt &b, const Pt &c, const Pt &d) {
    const i64 ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const i128 a2 = (i128)ax * ax + (i128)ay * ay, b2 = (i128)bx * bx + (i128)by * by, c2 = (i128)cx * cx + (i128)cy * cy;
    const i128 det = a2 * ((i128)bx * cy - (i128)by * cx) - b2 * ((i128)ax * cy - (i128)ay * cx) + c2 * ((i128)ax * by - (i128)ay * bx);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

// The same test in 64-bit arithmetic, exact while every coordinate difference is below 2^15 in magnitude (squared
// lengths < 2^31, cross products < 2^31, three products of < 2^62 cannot overflow once their signs are mixed; the
// caller enables it for |coordinates| <= 2^13, where the sum stays below 2^59).
inline int in_circle_small(const Pt &a, const Pt &b, const Pt &c, const Pt &d) {
    const i64 ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const i64 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const i64 det = a2 * (bx * cy - by * cx) - b2 * (ax * cy - ay * cx) + c2 * (ax * by - ay * bx);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

// Four cocircular points a, b, c (counter-clockwise), d: the sign the circle test takes when every point's lifted
// height x^2 + y^2 grows by an infinitesimal of its own, the larger the smaller the point's id.  The determinant is
// linear in the heights with coefficients +[b,c,d], -[a,c,d], +[a,b,d], -[a,b,c] (orientation determinants); the point
// with the smallest id decides, and its coefficient is not zero because no three points of a circle are collinear.
inline int in_circle_tie(const Pt &a, const Pt &b, const Pt &c, const Pt &d) {
    const int m = std::min(std::min(a.id, b.id), std::min(c.id, d.id));
    i64 o;
    if (m == a.id) o = orient(b, c, d);
    else if (m == b.id) o = -orient(a, c, d);
    else if (m == c.id) o = orient(a, b, d);
    else o = -orient(a, b, c);
    return o > 0 ? 1 : (o < 0 ? -1 : 0);
}

struct Tri { int v[3]; int n[3]; };      // n[i]: triangle across the edge opposite v[i], -1 on the hull

class Delaunay {
public:
    std::vector<Pt> p;                   // unique points in (x, y) order
    std::vector<Tri> t;
    bool small_ = false;                 // every |coordinate| <= 2^13: the 64-bit circle test is exact

    // p: the caller's points in any order, repeats allowed (the copy with the smallest id is kept)
    void build() {
        if (p.empty()) return;
        small_ = true;
        for (const Pt &q : p) small_ = small_ && q.x >= -8192 && q.x <= 8192 && q.y >= -8192 && q.y <= 8192;
        // Radial order around the input point nearest the middle of the bounding box (exact integer distances): the
        // points before any point q lie in the closed disc through q, so q lies outside their hull, and the hull grows
        // as a blob instead of a sliver, which keeps the number of flips per point small.  The same sort brings
        // repeated positions together.
        i64 x0 = p[0].x, x1 = p[0].x, y0 = p[0].y, y1 = p[0].y;
        for (const Pt &q : p) { x0 = std::min(x0, q.x); x1 = std::max(x1, q.x); y0 = std::min(y0, q.y); y1 = std::max(y1, q.y); }
        size_t c = 0;
        i64 best = -1;
        for (size_t i = 0; i < p.size(); ++i) {
            const i64 dx = 2 * p[i].x - (x0 + x1), dy = 2 * p[i].y - (y0 + y1), d2 = dx * dx + dy * dy;
            if (best < 0 || d2 < best) { best = d2; c = i; }
        }
        centre_ = p[c];
        for (Pt &q : p) q.d2 = (q.x - centre_.x) * (q.x - centre_.x) + (q.y - centre_.y) * (q.y - centre_.y);
        std::sort(p.begin(), p.end(), [](const Pt &a, const Pt &b) {
            return a.d2 != b.d2 ? a.d2 < b.d2 : (a.x != b.x ? a.x < b.x : (a.y != b.y ? a.y < b.y : a.id < b.id));
        });
        p.erase(std::unique(p.begin(), p.end(), [](const Pt &a, const Pt &b) { return a.x == b.x && a.y == b.y; }), p.end());
        const int n = (int)p.size();
        t.reserve(2 * (size_t)n);
        next_.assign(n, -1); prev_.assign(n, -1); hull_tri_.assign(n, -1); on_hull_.assign(n, 0);
        if (n < 3) return;
        int k = 2;                       // p[0..k) is the leading collinear run
        while (k < n && orient(p[0], p[1], p[k]) == 0) ++k;
        if (k == n) return;              // every point on one line: no triangle
        // the fan below needs the run in order along its line
        std::sort(p.begin(), p.begin() + k, [](const Pt &a, const Pt &b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
        seed(k);
        hash_.assign((size_t)1 << hash_bits(n), -1);
        for (int v = 0, u = k; v <= k; ++v, u = next_[u]) { on_hull_[u] = 1; hash_[hash_key(p[u])] = u; }
        for (int i = k + 1; i < n; ++i) add_outside(i, hull_hint(p[i]));
    }

    // Adds one more point to a mesh that has triangles.  A position that is already a vertex changes nothing.
    void insert(const Pt &q_in) {
        Pt q = q_in;
        q.d2 = 0;
        small_ = small_ && q.x >= -8192 && q.x <= 8192 && q.y >= -8192 && q.y <= 8192;
        int cur = last_tri_ < (int)t.size() ? last_tri_ : 0;
        for (;;) {                                   // visibility walk: cross any edge that has q strictly on its far side
            const Tri &x = t[cur];
            int i = 0;
            for (; i < 3; ++i)
                if (orient(p[x.v[(i + 1) % 3]], p[x.v[(i + 2) % 3]], q) < 0) break;
            if (i == 3) break;                       // q lies in the closed triangle
            if (x.n[i] < 0) {                        // left the hull through the edge v[i+1] -> v[i+2]
                const int hint = x.v[(i + 1) % 3];
                const int id = append(q);
                add_outside(id, hint);
                last_tri_ = (int)t.size() - 1;
                return;
            }
            cur = x.n[i];
        }
        int zero[3], nz = 0;
        for (int i = 0; i < 3; ++i)
            if (orient(p[t[cur].v[(i + 1) % 3]], p[t[cur].v[(i + 2) % 3]], q) == 0) zero[nz++] = i;
        last_tri_ = cur;
        if (nz >= 2) return;                         // on two edge lines of the triangle: one of its vertices
        const int id = append(q);
        if (nz == 0) split_triangle(cur, id); else split_edge(cur, zero[0], id);
        legalise();
    }

private:
    int last_tri_ = 0;

    int append(const Pt &q) {
        p.push_back(q);
        next_.push_back(-1); prev_.push_back(-1); hull_tri_.push_back(-1); on_hull_.push_back(0);
        return (int)p.size() - 1;
    }

    // q strictly inside triangle x = (v0, v1, v2): x becomes (v0, v1, q), two triangles are added.
    void split_triangle(int x, int q) {
        const int v0 = t[x].v[0], v1 = t[x].v[1], v2 = t[x].v[2], n0 = t[x].n[0], n1 = t[x].n[1], n2 = t[x].n[2];
        const int a = new_tri(v1, v2, q), b = new_tri(v2, v0, q);
        t[x].v[2] = q; t[x].n[0] = a; t[x].n[1] = b; t[x].n[2] = n2;
        t[a].n[0] = b; t[a].n[1] = x; t[a].n[2] = n0;
        t[b].n[0] = x; t[b].n[1] = a; t[b].n[2] = n1;
        if (n0 >= 0) t[n0].n[slot_of_neighbour(t[n0], x)] = a;
        if (n1 >= 0) t[n1].n[slot_of_neighbour(t[n1], x)] = b;
        note_hull_edges(x); note_hull_edges(a); note_hull_edges(b);
        stack_.push_back(std::make_pair(x, 2));
        stack_.push_back(std::make_pair(a, 2));
        stack_.push_back(std::make_pair(b, 2));
    }

    // q on the edge of triangle x opposite slot i, strictly between its end points.
    void split_edge(int x, int i, int q) {
        const int c = t[x].v[i], a = t[x].v[(i + 1) % 3], b = t[x].v[(i + 2) % 3];
        const int u = t[x].n[i], n_xa = t[x].n[(i + 1) % 3], n_xb = t[x].n[(i + 2) % 3];
        int d = -1, n_ub = -1, n_ua = -1;
        if (u >= 0) {
            const int j = slot_of_neighbour(t[u], x);      // u = (d, b, a)
            d = t[u].v[j]; n_ub = t[u].n[(j + 1) % 3]; n_ua = t[u].n[(j + 2) % 3];
        }
        const int x1 = new_tri(c, q, b);
        const int u1 = u >= 0 ? new_tri(d, q, a) : -1;
        Tri &X = t[x];
        X.v[0] = c; X.v[1] = a; X.v[2] = q; X.n[0] = u1; X.n[1] = x1; X.n[2] = n_xb;
        Tri &X1 = t[x1];
        X1.n[0] = u; X1.n[1] = n_xa; X1.n[2] = x;
        if (n_xa >= 0) t[n_xa].n[slot_of_neighbour(t[n_xa], x)] = x1;
        stack_.push_back(std::make_pair(x, 2));
        stack_.push_back(std::make_pair(x1, 1));
        if (u >= 0) {
            Tri &U = t[u];
            U.v[0] = d; U.v[1] = b; U.v[2] = q; U.n[0] = x1; U.n[1] = u1; U.n[2] = n_ua;
            Tri &U1 = t[u1];
            U1.n[0] = x; U1.n[1] = n_ub; U1.n[2] = u;
            if (n_ub >= 0) t[n_ub].n[slot_of_neighbour(t[n_ub], u)] = u1;
            stack_.push_back(std::make_pair(u, 2));
            stack_.push_back(std::make_pair(u1, 1));
            note_hull_edges(u); note_hull_edges(u1);
        } else {                                           // a hull edge a -> b becomes a -> q -> b
            next_[a] = q; prev_[q] = a; next_[q] = b; prev_[b] = q;
            on_hull_[q] = 1;
            hash_[hash_key(p[q])] = q;
        }
        note_hull_edges(x); note_hull_edges(x1);
    }

    std::vector<int> next_, prev_, hull_tri_;    // counter-clockwise hull; hull_tri_[u] owns edge u -> next_[u]
    std::vector<std::pair<int, int>> stack_;
    std::vector<char> on_hull_;
    std::vector<int> hash_;                      // hull vertices by direction from the centre: a starting hint only
    Pt centre_;

    static int hash_bits(int n) { int b = 4; while (((size_t)1 << (2 * b)) < (size_t)n) ++b; return b; }   // ~ sqrt(n) buckets
    size_t hash_key(const Pt &q) const {
        const double dx = (double)(q.x - centre_.x), dy = (double)(q.y - centre_.y);
        const double s = fabs(dx) + fabs(dy);
        if (s == 0.0) return 0;
        const double r = dx / s, a = (dy > 0.0 ? 3.0 - r : 1.0 + r) * 0.25;                  // monotone in the angle, in [0, 1]
        const size_t m = hash_.size();
        return std::min(m - 1, (size_t)(a * (double)m));
    }
    // A vertex still on the hull in (or after) the direction of q; any hull vertex is a correct hint, a near one a fast one.
    int hull_hint(const Pt &q) const {
        const size_t m = hash_.size(), k0 = hash_key(q);
        for (size_t j = 0; j < m; ++j) {
            const int v = hash_[(k0 + j) & (m - 1)];
            if (v >= 0 && on_hull_[v]) return v;
        }
        return 0;      // not reached: the hull always has vertices in the table
    }

    int new_tri(int a, int b, int c) {
        Tri x; x.v[0] = a; x.v[1] = b; x.v[2] = c; x.n[0] = x.n[1] = x.n[2] = -1;
        t.push_back(x);
        return (int)t.size() - 1;
    }
    static int slot_of_vertex(const Tri &x, int v) { return x.v[0] == v ? 0 : (x.v[1] == v ? 1 : 2); }
    static int slot_of_neighbour(const Tri &x, int nb) { return x.n[0] == nb ? 0 : (x.n[1] == nb ? 1 : 2); }

    // Fan from point k over the collinear run 0..k-1.
    void seed(int k) {
        const bool left = orient(p[0], p[k - 1], p[k]) > 0;      // k lies to the left of the run's direction
        int prev_tri = -1;
        for (int i = 0; i + 1 < k; ++i) {
            const int a = left ? i : i + 1, b = left ? i + 1 : i;    // (a, b, k) counter-clockwise
            const int cur = new_tri(a, b, k);
            if (prev_tri >= 0) {
                // the previous and current triangle share the edge (i, k)
                t[cur].n[slot_of_vertex(t[cur], i + 1)] = prev_tri;
                t[prev_tri].n[slot_of_vertex(t[prev_tri], i - 1)] = cur;
            }
            prev_tri = cur;
        }
        // hull, counter-clockwise: left: 0,1,..,k-1,k ; right: k-1,..,1,0,k
        std::vector<int> ring;
        if (left) for (int i = 0; i < k; ++i) ring.push_back(i);
        else for (int i = k - 1; i >= 0; --i) ring.push_back(i);
        ring.push_back(k);
        const int m = (int)ring.size();
        for (int i = 0; i < m; ++i) { next_[ring[i]] = ring[(i + 1) % m]; prev_[ring[(i + 1) % m]] = ring[i]; }
        for (int i = 0; i + 1 < k; ++i) hull_tri_[left ? i : i + 1] = i;          // run edges
        hull_tri_[ring[m - 2]] = left ? k - 2 : 0;                                 // last run point -> k
        hull_tri_[k] = left ? 0 : k - 2;                                           // k -> first ring point
    }

    bool sees(int u, int q) const { return orient(p[u], p[next_[u]], p[q]) < 0; }

    // Joins point q (outside the hull) to all hull edges it sees; `hint` is a hull vertex next to one.
    void add_outside(int q, int hint) {
        int first = -1;                   // a vertex whose hull edge q sees; the chain is then extended backwards
        if (sees(hint, q)) first = hint;
        else {                            // look both ways from the hint: the visible chain is near it
            int f = hint, b = hint;
            for (;;) {
                b = prev_[b];
                if (sees(b, q)) { first = b; break; }
                if (b == f) return;       // the whole hull was visited: q sees no edge (not expected for a point outside)
                f = next_[f];
                if (sees(f, q)) { first = f; break; }
                if (f == b) return;
            }
        }
        while (sees(prev_[first], q)) first = prev_[first];
        int u = first, prev_tri = -1, first_tri = -1;
        while (sees(u, q)) {
            if (u != first) on_hull_[u] = 0;                         // swallowed by the fan
            const int w = next_[u];
            const int old = hull_tri_[u];
            const int cur = new_tri(u, q, w);                        // counter-clockwise, see orient(u, w, q) < 0
            t[cur].n[1] = old;                                       // across (w, u), opposite q
            t[old].n[slot_of_hull_edge(old, u, w)] = cur;
            if (prev_tri >= 0) {                                     // shares edge (u, q) with the previous fan triangle
                t[cur].n[2] = prev_tri;                              // opposite w: edge (u, q)
                t[prev_tri].n[0] = cur;                              // previous = (u', q, u): opposite u' is edge (q, u)
            }
            stack_.push_back(std::make_pair(cur, 1));
            if (first_tri < 0) first_tri = cur;
            prev_tri = cur;
            u = w;
        }
        // hull: first -> q -> u; the fan's first triangle owns edge (first, q), its last owns (q, u)
        const int last = u;
        next_[first] = q; prev_[q] = first; next_[q] = last; prev_[last] = q;
        hull_tri_[first] = first_tri;
        hull_tri_[q] = prev_tri;
        on_hull_[q] = 1;
        hash_[hash_key(p[q])] = q;
        legalise();
    }

    // Slot of triangle x whose opposite edge is the hull edge u -> w.
    int slot_of_hull_edge(int x, int u, int w) const {
        for (int i = 0; i < 3; ++i)
            if (t[x].v[(i + 1) % 3] == u && t[x].v[(i + 2) % 3] == w) return i;
        return 0;
    }

    void note_hull_edges(int x) {
        for (int i = 0; i < 3; ++i)
            if (t[x].n[i] < 0) hull_tri_[t[x].v[(i + 1) % 3]] = x;
    }

    void legalise() {
        while (!stack_.empty()) {
            const int t1 = stack_.back().first, e1 = stack_.back().second;
            stack_.pop_back();
            const int t2 = t[t1].n[e1];
            if (t2 < 0) continue;
            const int e2 = slot_of_neighbour(t[t2], t1);
            const int p1 = t[t1].v[e1], a = t[t1].v[(e1 + 1) % 3], b = t[t1].v[(e1 + 2) % 3], p2 = t[t2].v[e2];
            int inside = small_ ? in_circle_small(p[p1], p[a], p[b], p[p2]) : in_circle(p[p1], p[a], p[b], p[p2]);
            if (inside == 0) inside = in_circle_tie(p[p1], p[a], p[b], p[p2]);
            if (inside <= 0) continue;
            // quad p1, a, p2, b (counter-clockwise): replace diagonal (a, b) by (p1, p2)
            const int n_p1a = t[t1].n[(e1 + 2) % 3], n_bp1 = t[t1].n[(e1 + 1) % 3];
            const int n_p2b = t[t2].n[(e2 + 2) % 3], n_ap2 = t[t2].n[(e2 + 1) % 3];
            Tri &x = t[t1], &y = t[t2];
            x.v[0] = p1; x.v[1] = a; x.v[2] = p2; x.n[0] = n_ap2; x.n[1] = t2; x.n[2] = n_p1a;
            y.v[0] = p1; y.v[1] = p2; y.v[2] = b; y.n[0] = n_p2b; y.n[1] = n_bp1; y.n[2] = t1;
            if (n_ap2 >= 0) t[n_ap2].n[slot_of_neighbour(t[n_ap2], t2)] = t1;
            if (n_bp1 >= 0) t[n_bp1].n[slot_of_neighbour(t[n_bp1], t1)] = t2;
            note_hull_edges(t1);
            note_hull_edges(t2);
            // p1 is the point being inserted: everything else was legal before, and edges that end in p1 stay legal, so
            // only the two edges now opposite p1 can have become illegal
            stack_.push_back(std::make_pair(t1, 0));
            stack_.push_back(std::make_pair(t2, 0));
        }
    }
};

struct TriOut { int v[3]; };

// The mesh's triangles as indices into the caller's point list (the first occurrence of a repeated position), each
// rotated so that its smallest index leads, sorted lexicographically: counting sort by the leading index (a point
// leads only a handful of triangles), then each small bucket by the other two.
void canonical_triangles(const Delaunay &d, int n, std::vector<TriOut> &out) {
    const size_t nt = d.t.size();
    std::vector<TriOut> tmp(nt);
    std::vector<int> start((size_t)n + 1, 0);
    for (size_t i = 0; i < nt; ++i) {
        const int v[3] = {d.p[d.t[i].v[0]].id, d.p[d.t[i].v[1]].id, d.p[d.t[i].v[2]].id};
        const int lead = v[0] < v[1] ? (v[0] < v[2] ? 0 : 2) : (v[1] < v[2] ? 1 : 2);
        for (int k = 0; k < 3; ++k) tmp[i].v[k] = v[(lead + k) % 3];
        ++start[(size_t)tmp[i].v[0] + 1];
    }
    for (int i = 0; i < n; ++i) start[(size_t)i + 1] += start[i];
    out.resize(nt);
    {
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (size_t i = 0; i < nt; ++i) out[(size_t)fill[tmp[i].v[0]]++] = tmp[i];
    }
    for (int i = 0; i < n; ++i)
        if (start[(size_t)i + 1] - start[i] > 1)
            std::sort(out.begin() + start[i], out.begin() + start[(size_t)i + 1], [](const TriOut &a, const TriOut &b) {
                return a.v[1] != b.v[1] ? a.v[1] < b.v[1] : a.v[2] < b.v[2];
            });
}

bool in_range(const int32_t *xy, int n) {
    for (int i = 0; i < 2 * n; ++i)
        if (xy[i] < -(1 << 23) || xy[i] > (1 << 23)) return false;
    return true;
}

void build_from(Delaunay &d, const int32_t *xy, int n) {
    d = Delaunay();
    d.p.resize(n);
    for (int i = 0; i < n; ++i) { d.p[i].x = xy[2 * i]; d.p[i].y = xy[2 * i + 1]; d.p[i].id = i; d.p[i].d2 = 0; }
    d.build();
}

// Triangulates; triangles come out in the canonical form above.
int triangulate(const int32_t *xy, int n, std::vector<TriOut> &out) {
    if (!in_range(xy, n)) return -1;
    Delaunay d;
    build_from(d, xy, n);
    canonical_triangles(d, n, out);
    return 0;
}

// Uploads the triangles' corners, fits the planes and rasterises the prior.
int prior_from_triangles(svx_ctx *ctx, const char *what, const int32_t *pts_xy_host, const std::vector<TriOut> &tris, const float *disparities,
                         int rows, int cols, float *prior) {
    const int nt = (int)tris.size();
    if (nt == 0) return SVX_OK;
    // device staging lives outside the workspace: the rasteriser reserves the workspace itself
    void *pin;
    int rc = svx_pinned_reserve(ctx, (size_t)nt * 6 * sizeof(int32_t), &pin);
    if (rc) return rc;
    int32_t *h = (int32_t *)pin;
    for (int i = 0; i < nt; ++i)
        for (int k = 0; k < 3; ++k) {
            h[6 * i + 2 * k] = pts_xy_host[2 * tris[i].v[k]];
            h[6 * i + 2 * k + 1] = pts_xy_host[2 * tris[i].v[k] + 1];
        }
    // triangle corners and planes live in a block of the context that is kept between calls
    const size_t xy_bytes = (size_t)nt * 6 * sizeof(int32_t), abc_bytes = (size_t)nt * 3 * sizeof(float);
    rc = svx_block_reserve(ctx, &ctx->tri, &ctx->tri_bytes, xy_bytes + abc_bytes, what);
    if (rc) return rc;
    int32_t *d_xy = (int32_t *)ctx->tri;
    float *d_abc = (float *)((char *)ctx->tri + xy_bytes);
    if (hipMemcpyAsync(d_xy, h, xy_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return svx_fail(ctx, SVX_ERR_HIP, "%s: upload failed", what);
    rc = svx_pinned_mark_read(ctx);
    if (!rc) rc = svx_plane_fit(ctx, d_xy, nt, disparities, rows, cols, d_abc);
    if (!rc) rc = svx_plane_prior_raster(ctx, d_xy, d_abc, nt, rows, cols, prior);
    return rc;
}

}  // namespace

SVX_API int svx_delaunay(svx_ctx *ctx, const int32_t *pts_xy_host, int n_pts, int32_t *tri_idx_host, int capacity, int *n_tri) {
    // host-only: ctx may be null (then no message is recorded)
    SVX_REQUIRE(ctx, n_pts >= 0 && capacity >= 0 && n_tri, SVX_ERR_INVALID_ARG, "svx_delaunay: bad count or null result");
    SVX_REQUIRE(ctx, n_pts == 0 || pts_xy_host, SVX_ERR_INVALID_ARG, "svx_delaunay: null point list");
    SVX_REQUIRE(ctx, capacity == 0 || tri_idx_host, SVX_ERR_INVALID_ARG, "svx_delaunay: null triangle buffer");
    std::vector<TriOut> tris;
    if (triangulate(pts_xy_host, n_pts, tris) != 0)
        return svx_fail(ctx, SVX_ERR_UNSUPPORTED, "svx_delaunay: coordinates must lie within +-2^23");
    *n_tri = (int)tris.size();
    const int m = std::min((int)tris.size(), capacity);
    for (int i = 0; i < m; ++i)
        for (int k = 0; k < 3; ++k) tri_idx_host[3 * i + k] = tris[i].v[k];
    return SVX_OK;
}

// Triangulates the support points, fits the triangles' disparity planes and rasterises the plane prior.
SVX_API int svx_mesh_plane_prior(svx_ctx *ctx, const int32_t *pts_xy_host, int n_pts, const float *disparities, int rows, int cols,
                                 float *prior, int *n_tri_out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, n_pts >= 0, SVX_ERR_INVALID_ARG, "svx_mesh_plane_prior: negative point count");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_mesh_plane_prior: empty image");
    SVX_REQUIRE(ctx, disparities && prior && (n_pts == 0 || pts_xy_host), SVX_ERR_INVALID_ARG, "svx_mesh_plane_prior: null pointer");
    for (int i = 0; i < n_pts; ++i)
        SVX_REQUIRE(ctx, pts_xy_host[2 * i] >= 0 && pts_xy_host[2 * i] < cols && pts_xy_host[2 * i + 1] >= 0 && pts_xy_host[2 * i + 1] < rows,
                    SVX_ERR_INVALID_ARG, "svx_mesh_plane_prior: support point outside the image");
    std::vector<TriOut> tris;
    if (triangulate(pts_xy_host, n_pts, tris) != 0) return svx_fail(ctx, SVX_ERR_UNSUPPORTED, "svx_mesh_plane_prior: coordinate range");
    if (n_tri_out) *n_tri_out = (int)tris.size();
    return prior_from_triangles(ctx, "svx_mesh_plane_prior", pts_xy_host, tris, disparities, rows, cols, prior);
}

// ---- the mesh as an object that grows: Fade_2D::insert / getTrianglePointers, highPerfStereoLib.cpp:434-445, :457 ----
struct svx_mesh {
    Delaunay d;
    std::vector<int32_t> xy;      // every point handed in, in order: the index space of the triangles
};

SVX_API int svx_mesh_create(svx_mesh **out) {
    if (!out) return SVX_ERR_INVALID_ARG;
    *out = new (std::nothrow) svx_mesh();
    return *out ? SVX_OK : SVX_ERR_NOMEM;
}

SVX_API void svx_mesh_destroy(svx_mesh *mesh) { delete mesh; }

SVX_API int svx_mesh_insert(svx_mesh *mesh, const int32_t *pts_xy_host, int n_pts) {
    if (!mesh || n_pts < 0 || (n_pts > 0 && !pts_xy_host)) return SVX_ERR_INVALID_ARG;
    if (!in_range(pts_xy_host, n_pts)) return SVX_ERR_UNSUPPORTED;
    const int before = (int)(mesh->xy.size() / 2);
    mesh->xy.insert(mesh->xy.end(), pts_xy_host, pts_xy_host + 2 * (size_t)n_pts);
    if (mesh->d.t.empty()) {           // nothing to insert into yet (fewer than three points, or all on one line)
        build_from(mesh->d, mesh->xy.data(), before + n_pts);
        return SVX_OK;
    }
    for (int i = 0; i < n_pts; ++i) {
        Pt q;
        q.x = pts_xy_host[2 * i]; q.y = pts_xy_host[2 * i + 1]; q.id = before + i; q.d2 = 0;
        mesh->d.insert(q);
    }
    return SVX_OK;
}

SVX_API int svx_mesh_counts(const svx_mesh *mesh, int *n_points, int *n_triangles) {
    if (!mesh) return SVX_ERR_INVALID_ARG;
    if (n_points) *n_points = (int)(mesh->xy.size() / 2);
    if (n_triangles) *n_triangles = (int)mesh->d.t.size();
    return SVX_OK;
}

SVX_API int svx_mesh_triangles(const svx_mesh *mesh, int32_t *tri_idx_host, int capacity, int *n_tri) {
    if (!mesh || capacity < 0 || !n_tri || (capacity > 0 && !tri_idx_host)) return SVX_ERR_INVALID_ARG;
    std::vector<TriOut> tris;
    canonical_triangles(mesh->d, (int)(mesh->xy.size() / 2), tris);
    *n_tri = (int)tris.size();
    const int m = std::min((int)tris.size(), capacity);
    for (int i = 0; i < m; ++i)
        for (int k = 0; k < 3; ++k) tri_idx_host[3 * i + k] = tris[i].v[k];
    return SVX_OK;
}

SVX_API int svx_mesh_prior(svx_ctx *ctx, const svx_mesh *mesh, const float *disparities, int rows, int cols, float *prior, int *n_tri_out) {
    int rc = svx_enter(ctx);
    if (rc) return rc;
    SVX_REQUIRE(ctx, mesh && disparities && prior, SVX_ERR_INVALID_ARG, "svx_mesh_prior: null pointer");
    SVX_REQUIRE(ctx, rows > 0 && cols > 0, SVX_ERR_EMPTY, "svx_mesh_prior: empty image");
    const int n_pts = (int)(mesh->xy.size() / 2);
    for (int i = 0; i < n_pts; ++i)
        SVX_REQUIRE(ctx, mesh->xy[2 * i] >= 0 && mesh->xy[2 * i] < cols && mesh->xy[2 * i + 1] >= 0 && mesh->xy[2 * i + 1] < rows,
                    SVX_ERR_INVALID_ARG, "svx_mesh_prior: mesh vertex outside the image");
    std::vector<TriOut> tris;
    canonical_triangles(mesh->d, n_pts, tris);
    if (n_tri_out) *n_tri_out = (int)tris.size();
    return prior_from_triangles(ctx, "svx_mesh_prior", mesh->xy.data(), tris, disparities, rows, cols, prior);
}
