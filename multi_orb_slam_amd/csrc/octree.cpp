// octree.cpp -- host quadtree keypoint distribution (reference src/ORBextractor.cc:482-764, DistributeOctTree +
// ExtractorNode::DivideNode), array-based: nodes live in a pool, the std::list order of the reference is kept with
// prev/next indices, a node's keypoints are one contiguous slice of an index buffer that is stably 4-way partitioned
// in place when the node splits.
//
// Determinism: the reference sorts pair<int, ExtractorNode*> (src/ORBextractor.cc:685), i.e. ties between equally
// populated nodes are broken by heap address.  Canonical replacement: creation order -- among nodes
// of equal size the most recently created one is expanded first.  Pool index == creation order here.
#include "octree.h"

#include <algorithm>
#include <cmath>
#include <utility>

namespace morb {

namespace {

struct QNode {
    int ulx, uly, brx, bry;  // UL and BR corners; UR = (brx, uly), BL = (ulx, bry)
    int begin, end;          // slice of Octree::idx
    int prev, next;          // list links (-1 = none)
    bool no_more;
};

struct Octree {
    const int* x; const int* y; const int* resp;
    std::vector<QNode> pool;
    std::vector<int> idx, tmp;
    int head = -1, tail = -1, count = 0;

    void push_front(int n) {
        pool[n].prev = -1; pool[n].next = head;
        if (head >= 0) pool[head].prev = n; else tail = n;
        head = n; ++count;
    }
    void push_back(int n) {
        pool[n].next = -1; pool[n].prev = tail;
        if (tail >= 0) pool[tail].next = n; else head = n;
        tail = n; ++count;
    }
    void erase(int n) {
        const int p = pool[n].prev, q = pool[n].next;
        if (p >= 0) pool[p].next = q; else head = q;
        if (q >= 0) pool[q].prev = p; else tail = p;
        --count;
    }

    // Splits node n; children with >= 1 point are pushed to the list front in the order n1..n4 (so n4 ends up first);
    // children with > 1 point are appended to `expandable` as (size, node).  The parent is NOT erased here.
    void divide(int n, std::vector<std::pair<int, int>>* expandable, int* n_to_expand) {
        const QNode P = pool[n];
        const int halfX = (int)std::ceil((float)(P.brx - P.ulx) / 2);
        const int halfY = (int)std::ceil((float)(P.bry - P.uly) / 2);
        const int midx = P.ulx + halfX, midy = P.uly + halfY;
        int cnt[4] = {0, 0, 0, 0};
        for (int k = P.begin; k < P.end; ++k) {
            const int i = idx[k];
            const int c = (x[i] < midx ? 0 : 1) + (y[i] < midy ? 0 : 2);  // n1=0 (UL), n2=1 (UR), n3=2 (BL), n4=3 (BR)
            ++cnt[c];
        }
        int off[4] = {P.begin, P.begin + cnt[0], P.begin + cnt[0] + cnt[1], P.begin + cnt[0] + cnt[1] + cnt[2]};
        int cur[4] = {off[0], off[1], off[2], off[3]};
        for (int k = P.begin; k < P.end; ++k) {
            const int i = idx[k];
            const int c = (x[i] < midx ? 0 : 1) + (y[i] < midy ? 0 : 2);
            tmp[cur[c]++] = i;
        }
        std::copy(tmp.begin() + P.begin, tmp.begin() + P.end, idx.begin() + P.begin);
        const int cx0[4] = {P.ulx, midx, P.ulx, midx}, cy0[4] = {P.uly, P.uly, midy, midy};
        const int cx1[4] = {midx, P.brx, midx, P.brx}, cy1[4] = {midy, midy, P.bry, P.bry};
        for (int c = 0; c < 4; ++c) {
            if (cnt[c] == 0) continue;
            QNode ch;
            ch.ulx = cx0[c]; ch.uly = cy0[c]; ch.brx = cx1[c]; ch.bry = cy1[c];
            ch.begin = off[c]; ch.end = off[c] + cnt[c];
            ch.prev = ch.next = -1;
            ch.no_more = cnt[c] == 1;
            pool.push_back(ch);
            const int id = (int)pool.size() - 1;
            push_front(id);
            if (cnt[c] > 1) {
                if (n_to_expand) ++*n_to_expand;
                expandable->push_back(std::make_pair(cnt[c], id));
            }
        }
    }
};

}  // namespace

void distribute_octree(const int* x, const int* y, const int* resp, int n, int width, int height, int N,
                       std::vector<int>& selected) {
    selected.clear();
    if (n <= 0 || width <= 0 || height <= 0) return;
    Octree T;
    T.x = x; T.y = y; T.resp = resp;
    T.idx.resize(n); T.tmp.resize(n);
    T.pool.reserve((size_t)4 * std::max(N, 16) + 64);

    // roots: nIni = round(width/height) vertical strips (src/ORBextractor.cc:544-565)
    const int nIni = std::max(1, (int)std::round((float)width / (float)height));
    const float hX = (float)width / (float)nIni;
    std::vector<int> root_cnt(nIni, 0), root_of(n);
    for (int i = 0; i < n; ++i) {
        int r = (int)((float)x[i] / hX);
        r = std::min(std::max(r, 0), nIni - 1);
        root_of[i] = r; ++root_cnt[r];
    }
    std::vector<int> root_off(nIni + 1, 0);
    for (int r = 0; r < nIni; ++r) root_off[r + 1] = root_off[r] + root_cnt[r];
    {
        std::vector<int> cur(root_off.begin(), root_off.end() - 1);
        for (int i = 0; i < n; ++i) T.idx[cur[root_of[i]]++] = i;  // stable: input order kept inside a root
    }
    for (int r = 0; r < nIni; ++r) {
        if (root_cnt[r] == 0) continue;  // empty roots are erased right away (:575-585)
        QNode q;
        q.ulx = (int)(hX * (float)r); q.uly = 0;
        q.brx = (int)(hX * (float)(r + 1)); q.bry = height;
        q.begin = root_off[r]; q.end = root_off[r + 1];
        q.prev = q.next = -1;
        q.no_more = root_cnt[r] == 1;
        T.pool.push_back(q);
        T.push_back((int)T.pool.size() - 1);
    }

    std::vector<std::pair<int, int>> expandable, prev_exp;
    bool finish = false;
    while (!finish) {
        const int prev_size = T.count;
        int n_to_expand = 0;
        expandable.clear();
        // full pass over the list: split every node that still holds more than one point (:600-662)
        for (int it = T.head; it >= 0;) {
            const int nxt = T.pool[it].next;
            if (!T.pool[it].no_more) {
                T.divide(it, &expandable, &n_to_expand);
                T.erase(it);
            }
            it = nxt;
        }
        if (T.count >= N || T.count == prev_size) {
            finish = true;
        } else if (T.count + n_to_expand * 3 > N) {
            // careful phase: expand the most populated nodes first and stop the moment N is reached (:674-738)
            while (!finish) {
                const int ps = T.count;
                prev_exp.swap(expandable);
                expandable.clear();
                std::sort(prev_exp.begin(), prev_exp.end());  // (size, creation index) ascending
                for (int j = (int)prev_exp.size() - 1; j >= 0; --j) {
                    const int node = prev_exp[j].second;
                    T.divide(node, &expandable, nullptr);
                    T.erase(node);
                    if (T.count >= N) break;
                }
                if (T.count >= N || T.count == ps) finish = true;
            }
        }
    }
    // best response per node, first one wins a tie (:742-763); output in list order
    selected.reserve(T.count);
    for (int it = T.head; it >= 0; it = T.pool[it].next) {
        const QNode& q = T.pool[it];
        int best = T.idx[q.begin];
        for (int k = q.begin + 1; k < q.end; ++k)
            if (resp[T.idx[k]] > resp[best]) best = T.idx[k];
        selected.push_back(best);
    }
}

}  // namespace morb
