// AddressSanitizer + UndefinedBehaviorSanitizer driver of the host-only code: the CPU oracle (oracle/orb_oracle.cpp, bow_oracle.cpp)
// and the product's host quadtree (multi_orb_slam_amd/csrc/octree.cpp: the fallback of the device quadtree).  Runs the whole extraction
// on images of many sizes, pitches and kinds (rectangles, noise, flat, tiny), the two quadtree implementations against each other on
// every level's candidates, and the grid / projection-search / brute-force restatements on random two-camera frames.  Exit code 0 and
// no sanitizer report = clean (tests/test_sanitizers.py).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../multi_orb_slam_amd/csrc/octree.h"

struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct orc_query { float u, v, radius, ur; int min_level, max_level, cam, blocks; float angle; uint8_t desc[32]; };
struct orc_frame {
    int n_total, n_cams;
    const float* un_x; const float* un_y; const int* octave; const float* angle; const float* uright;
    const int* cam_of; const int* local_of;
    const uint8_t* const* desc;
    float minX, minY, maxX, maxY;
};
extern "C" {
int orc_keypoint_size();
void orc_level_sizes(int W, int H, float scaleFactor, int nlevels, int* w, int* h);
void orc_pyramid(const uint8_t* img, int W, int H, int stride, float scaleFactor, int nlevels, uint8_t* out);
int orc_cell_candidates(const uint8_t* img, int w, int h, int iniTh, int minTh, KeyPoint* out, int cap);
int orc_distribute_octree(const KeyPoint* in, int n_in, int minX, int maxX, int minY, int maxY, int N, KeyPoint* out, int cap);
int orc_extract(const uint8_t* img, int W, int H, int stride, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                KeyPoint* kps_out, uint8_t* desc_out, int cap);
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orc_bf_top2(const uint8_t* q, int nq, const uint8_t* r, int nr, int* best_idx, int* best_dist, int* second_dist);
void orc_hamming_matrix(const uint8_t* q, int nq, const uint8_t* r, int nr, uint16_t* out);
void orc_three_maxima(const int* sizes, int L, int* ind);
void orc_grid_csr(const orc_frame* f, int* cell_start, int* items);
int orc_features_in_area(const orc_frame* f, int cam, float x, float y, float r, int minLevel, int maxLevel, int* out, int cap);
int orc_search_by_projection_frames(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied, int th_high, int check_ori,
                                    int* match_of_feature);
int orc_search_by_projection_points(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied, float nnratio, int th_high,
                                    int* match_of_feature);
void orc_project_best(const orc_frame* cur, const orc_query* q, int nq, const uint8_t* occupied, int gate, const float* inv_sigma2,
                      int* best_idx, int* best_dist);
void orc_undistort_points(const float* calib, const float* x, const float* y, int n, float* ux, float* uy);
void orc_image_bounds(const float* calib, int cols, int rows, float* out4);
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return (uint32_t)(g_state >> 16); }

static std::vector<uint8_t> make_image(int w, int h, int stride, int kind) {
    std::vector<uint8_t> img((size_t)stride * h, 0xCD);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img[(size_t)y * stride + x] = kind == 2 ? 77 : kind == 1 ? (uint8_t)rnd() : (uint8_t)(128 + (int)(rnd() % 13) - 6);
    if (kind == 0) {
        const int n = 8 + w * h / 800;
        for (int k = 0; k < n; ++k) {
            const int x0 = (int)(rnd() % (unsigned)(w + 60)) - 30, y0 = (int)(rnd() % (unsigned)(h + 60)) - 30, rw = 4 + (int)(rnd() % 61), rh = 4 + (int)(rnd() % 61);
            const uint8_t g = (uint8_t)rnd();
            for (int y = y0 < 0 ? 0 : y0; y < y0 + rh && y < h; ++y)
                for (int x = x0 < 0 ? 0 : x0; x < x0 + rw && x < w; ++x) img[(size_t)y * stride + x] = g;
        }
    }
    return img;
}

int main() {
    int bad = 0;
    if (orc_keypoint_size() != (int)sizeof(KeyPoint)) return 2;
    const struct { int w, h, stride, kind, nf; } cases[] = {
        {640, 480, 640, 0, 1000}, {641, 479, 700, 0, 500}, {320, 240, 320, 1, 300}, {97, 83, 128, 0, 200}, {64, 48, 64, 0, 100},
        {40, 40, 40, 1, 50}, {400, 300, 400, 2, 400}, {752, 480, 768, 1, 1200}, {1280, 720, 1280, 0, 2000}};
    for (const auto& c : cases) {
        std::vector<uint8_t> img = make_image(c.w, c.h, c.stride, c.kind);
        const int cap = c.nf + 64;
        std::vector<KeyPoint> kps(cap);
        std::vector<uint8_t> desc((size_t)cap * 32);
        const int n = orc_extract(img.data(), c.w, c.h, c.stride, c.nf, 1.2f, 8, 20, 7, kps.data(), desc.data(), cap);
        if (n == INT32_MIN) continue;   // (a level so small that width / height rounds to zero quadtree roots: undefined in the reference, refused by the oracle)
        if (n < 0 || n > cap) { std::printf("extract %dx%d: %d\n", c.w, c.h, n); ++bad; }
        if (c.kind == 2 && n != 0) ++bad;
        if (n > 8) {   // an output buffer that is too small is reported, never overrun
            const int small = orc_extract(img.data(), c.w, c.h, c.stride, c.nf, 1.2f, 8, 20, 7, kps.data(), desc.data(), 8);
            if (small != -n) { std::printf("extract cap 8: %d (n %d)\n", small, n); ++bad; }
        }
        // every level's candidates through both quadtrees
        int lw[8], lh[8];
        orc_level_sizes(c.w, c.h, 1.2f, 8, lw, lh);
        size_t total = 0;
        for (int l = 0; l < 8; ++l) total += (size_t)lw[l] * lh[l];
        std::vector<uint8_t> pyr(total);
        orc_pyramid(img.data(), c.w, c.h, c.stride, 1.2f, 8, pyr.data());
        size_t off = 0;
        for (int l = 0; l < 8; ++l) {
            const uint8_t* L = pyr.data() + off;
            off += (size_t)lw[l] * lh[l];
            if (lw[l] < 40 || lh[l] < 40) continue;   // (below 2 * 16 + 7 nothing is scored)
            std::vector<KeyPoint> cand(400000);
            const int nc = orc_cell_candidates(L, lw[l], lh[l], 20, 7, cand.data(), (int)cand.size());
            if (nc < 0 || nc > (int)cand.size()) { ++bad; continue; }
            cand.resize(nc);
            const int quota = 10 + c.nf / (l + 2);
            const int minX = 16, maxX = lw[l] - 16, minY = 16, maxY = lh[l] - 16;
            if (maxX - minX < 2 || maxY - minY < 2) continue;
            std::vector<KeyPoint> sel(nc + 8);
            const int ns = orc_distribute_octree(cand.data(), nc, minX, maxX, minY, maxY, quota, sel.data(), (int)sel.size());
            std::vector<int> x(nc), y(nc), r(nc), picked;
            for (int i = 0; i < nc; ++i) { x[i] = (int)cand[i].x; y[i] = (int)cand[i].y; r[i] = (int)cand[i].response; }
            if (ns == INT32_MIN) continue;   // (a geometry the reference itself leaves undefined: width / height rounds to 0 roots)
            morb::distribute_octree(x.data(), y.data(), r.data(), nc, maxX - minX, maxY - minY, quota, picked);
            if ((int)picked.size() != ns) { std::printf("quadtree sizes %dx%d level %d: %d vs %d\n", c.w, c.h, l, (int)picked.size(), ns); ++bad; continue; }
            for (int i = 0; i < ns; ++i)
                if ((int)sel[i].x != x[picked[i]] || (int)sel[i].y != y[picked[i]] || (int)sel[i].response != r[picked[i]]) { ++bad; break; }
        }
    }
    // ---- matcher restatements on random frames (two cameras, features also outside the image bounds and on the grid's last half cell)
    for (int rep = 0; rep < 6; ++rep) {
        const int n0 = 50 + (int)(rnd() % 700), n1 = (int)(rnd() % 500), n = n0 + n1, W = 640, H = 480;
        std::vector<float> ux(n), uy(n), ang(n), ur(n);
        std::vector<int> oct(n), cam(n), loc(n);
        std::vector<uint8_t> d0((size_t)n0 * 32 + 1), d1((size_t)n1 * 32 + 1);
        for (auto& b : d0) b = (uint8_t)rnd();
        for (auto& b : d1) b = (uint8_t)rnd();
        for (int g = 0; g < n; ++g) {
            ux[g] = (float)(rnd() % (unsigned)(W + 20)) - 10.f + 0.25f * (float)(rnd() % 4);
            uy[g] = (float)(rnd() % (unsigned)(H + 20)) - 10.f;
            ang[g] = (float)(rnd() % 36000) / 100.f; ur[g] = (rnd() & 1) ? -1.f : ux[g] - 5.f; oct[g] = (int)(rnd() % 8);
            cam[g] = g < n0 ? 0 : 1; loc[g] = g < n0 ? g : g - n0;
        }
        const uint8_t* dptr[2] = {d0.data(), d1.data()};
        orc_frame F{n, 2, ux.data(), uy.data(), oct.data(), ang.data(), ur.data(), cam.data(), loc.data(), dptr, 0.f, 0.f, (float)W, (float)H};
        std::vector<int> cell_start(2 * 3072 + 1), items(n);
        orc_grid_csr(&F, cell_start.data(), items.data());
        if (cell_start[2 * 3072] > n) ++bad;
        const int nq = 40 + (int)(rnd() % 600);
        std::vector<orc_query> q(nq);
        for (int i = 0; i < nq; ++i) {
            const int g = (int)(rnd() % (unsigned)n);
            q[i].u = ux[g] + (float)(rnd() % 9) - 4.f; q[i].v = uy[g] + (float)(rnd() % 9) - 4.f;
            q[i].radius = 5.f + (float)(rnd() % 60); q[i].ur = q[i].u - 4.f;
            q[i].min_level = (int)(rnd() % 9) - 1; q[i].max_level = (int)(rnd() % 9) - 1; q[i].cam = cam[g]; q[i].blocks = (int)(rnd() & 1);
            q[i].angle = ang[g]; std::memcpy(q[i].desc, dptr[cam[g]] + (size_t)loc[g] * 32, 32);
            if (i % 3 == 0) q[i].desc[rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8));
            if (i % 17 == 0) { q[i].u = -500.f; q[i].v = 1e6f; }       // far outside: empty windows
        }
        std::vector<int> out(n + 16), mof(n), bi(nq), bd(nq);
        for (int i = 0; i < nq; i += 7) (void)orc_features_in_area(&F, q[i].cam, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, out.data(), (int)out.size());
        (void)orc_features_in_area(&F, 0, 320.f, 240.f, 1e4f, -1, -1, out.data(), 4);      // more hits than room: counted, not written
        std::vector<uint8_t> occ(n);
        for (auto& b : occ) b = (uint8_t)(rnd() % 5 == 0);
        for (int ori = 0; ori < 2; ++ori) {
            (void)orc_search_by_projection_frames(&F, q.data(), nq, ori ? occ.data() : nullptr, 100, ori, mof.data());
            (void)orc_search_by_projection_points(&F, q.data(), nq, ori ? occ.data() : nullptr, 0.8f, 100, mof.data());
        }
        float inv_s2[8]; for (int l = 0; l < 8; ++l) inv_s2[l] = 1.f / (1.f + (float)l);
        for (int gate = 0; gate < 3; ++gate) orc_project_best(&F, q.data(), nq, occ.data(), gate, inv_s2, bi.data(), bd.data());
        (void)orc_search_by_projection_frames(&F, q.data(), 0, nullptr, 100, 1, mof.data());   // no queries
        std::vector<int> b_i(n0), b_d(n0), s_d(n0);
        orc_bf_top2(d0.data(), n0, d1.data(), n1, b_i.data(), b_d.data(), s_d.data());          // (n1 may be 0)
        std::vector<uint16_t> mat((size_t)n0 * (n1 ? n1 : 1));
        orc_hamming_matrix(d0.data(), n0, d1.data(), n1, mat.data());
        if (n1 && mat[0] != (uint16_t)orc_descriptor_distance(d0.data(), d1.data())) ++bad;
        if (orc_descriptor_distance(d0.data() + 1, d0.data() + 1) != 0) ++bad;                  // (rows need no alignment)
    }
    int sizes[30] = {0}, ind[3];
    orc_three_maxima(sizes, 30, ind);
    for (int k = 0; k < 30; ++k) sizes[k] = (int)(rnd() % 50);
    orc_three_maxima(sizes, 30, ind);
    const float calib[9] = {520.9f, 521.0f, 325.1f, 249.7f, 0.26f, -0.95f, -0.005f, 0.002f, 1.16f};
    float px[4] = {0.f, 640.f, 0.f, 640.f}, py[4] = {0.f, 0.f, 480.f, 480.f}, ox[4], oy[4], bounds[4];
    orc_undistort_points(calib, px, py, 4, ox, oy);
    orc_image_bounds(calib, 640, 480, bounds);
    if (!(bounds[2] > bounds[0] && bounds[3] > bounds[1])) ++bad;
    std::printf("san_oracle: %d failures\n", bad);
    return bad ? 1 : 0;
}
