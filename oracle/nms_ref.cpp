// CPU restatement of tf.image.non_max_suppression as TF 1.13 executes it
// (NonMaxSuppressionV3, tensorflow/core/kernels/non_max_suppression_op.cc,
// tensorflow==1.13.x -- third-party dependency of the reference, not vendored;
// restated from the published algorithm).  TEST INFRASTRUCTURE ONLY.
//
// Call sites in the reference: SSD300.py:179 (per-class inference NMS) and
// SSD300.py:431 (hard-negative mining).
//
// Semantics mirrored:
//  * candidates = every box with score > score_threshold, pushed in index order
//    into std::priority_queue<Candidate, std::deque<Candidate>, cmp(score <)>
//    -> ties are resolved by libstdc++'s heap order (deterministic, not stable);
//  * pop best; compare with already selected boxes (newest first); suppress iff
//    IoU > iou_threshold (strict); stop at max_output;
//  * IoU: coordinate-order agnostic (min/max of the two y's / x's), 0 if either
//    area <= 0, all float32, no FMA contraction (build with -ffp-contract=off).
#include <algorithm>
#include <deque>
#include <queue>
#include <vector>

namespace {
struct Candidate { int box_index; float score; };

inline float iou_ref(const float* boxes, int i, int j) {
    const float ymin_i = std::min<float>(boxes[i * 4 + 0], boxes[i * 4 + 2]);
    const float xmin_i = std::min<float>(boxes[i * 4 + 1], boxes[i * 4 + 3]);
    const float ymax_i = std::max<float>(boxes[i * 4 + 0], boxes[i * 4 + 2]);
    const float xmax_i = std::max<float>(boxes[i * 4 + 1], boxes[i * 4 + 3]);
    const float ymin_j = std::min<float>(boxes[j * 4 + 0], boxes[j * 4 + 2]);
    const float xmin_j = std::min<float>(boxes[j * 4 + 1], boxes[j * 4 + 3]);
    const float ymax_j = std::max<float>(boxes[j * 4 + 0], boxes[j * 4 + 2]);
    const float xmax_j = std::max<float>(boxes[j * 4 + 1], boxes[j * 4 + 3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0 || area_j <= 0) return 0.0f;
    const float iymin = std::max<float>(ymin_i, ymin_j);
    const float ixmin = std::max<float>(xmin_i, xmin_j);
    const float iymax = std::min<float>(ymax_i, ymax_j);
    const float ixmax = std::min<float>(xmax_i, xmax_j);
    const float inter = std::max<float>(iymax - iymin, 0.0f) * std::max<float>(ixmax - ixmin, 0.0f);
    return inter / (area_i + area_j - inter);
}
}  // namespace

extern "C" int nms_ref_v3(const float* boxes, const float* scores, int n, int max_output,
                          float iou_threshold, float score_threshold, int* selected_out) {
    auto cmp = [](const Candidate a, const Candidate b) { return a.score < b.score; };
    std::priority_queue<Candidate, std::deque<Candidate>, decltype(cmp)> pq(cmp);
    for (int i = 0; i < n; ++i)
        if (scores[i] > score_threshold) pq.push(Candidate{i, scores[i]});
    std::vector<int> selected;
    while ((int)selected.size() < max_output && !pq.empty()) {
        Candidate next = pq.top();
        pq.pop();
        bool keep = true;
        for (int j = (int)selected.size() - 1; j >= 0; --j) {
            if (iou_ref(boxes, next.box_index, selected[j]) > iou_threshold) { keep = false; break; }
        }
        if (keep) selected.push_back(next.box_index);
    }
    for (size_t i = 0; i < selected.size(); ++i) selected_out[i] = selected[i];
    return (int)selected.size();
}
