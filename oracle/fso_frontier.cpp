/*
 * fso_frontier.cpp — CPU ORACLE (test infrastructure, never linked into the product): restatement of the reference's
 * frontier detection + clustering, the producer of the candidate list.
 *
 * Follows, statement for statement:
 *   DEP/src/FrontierSearch.cpp:21-96    FrontierSearch::searchFrom     (outer 4-connected BFS over cells < LETHAL_OBSTACLE)
 *   DEP/src/FrontierSearch.cpp:98-216   FrontierSearch::buildNewFrontier (8-connected BFS over frontier cells, split at
 *                                       max_frontier_cluster_size, goal point = angular median, FRONTIER_POINT_MEDIAN build:
 *                                       DEP/CMakeLists.txt:64)
 *   DEP/src/FrontierSearch.cpp:218-249  isNewFrontierCell
 *   DEP/include/frontier_exploration/FrontierSearch.hpp:84-127,129-142,156-181  getCentroidOfCells, isLethal/isUnknown/isFree,
 *                                       SortByMedianFunctor
 *   DEP/src/Helpers.cpp:185-255,285-329 nhood4, nhood8, nearestFreeCell
 *   DEP/include/frontier_exploration/util/GeometryUtils.hpp:102-105  distanceBetweenPoints
 * (DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/).  nav2_costmap_2d::Costmap2D accessors as in
 * SURVEY.md App. B.  C++ (g++, libstdc++) because the goal point depends on std::sort applied to a comparator that is
 * not a strict weak order: the result is whatever libstdc++'s introsort makes of it, so the same library is used.
 *
 * Parity unpinned: the reference holds no fixture for this function; pinned by hand-built maps in
 * tests/test_oracle_frontier.py.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace {

struct Map {
    const uint8_t *map;
    unsigned int size_x, size_y;
    double origin_x, origin_y, resolution;
    unsigned int getIndex(unsigned int mx, unsigned int my) const { return my * size_x + mx; }
    void indexToCells(unsigned int index, unsigned int &mx, unsigned int &my) const { my = index / size_x; mx = index - (my * size_x); }
    void mapToWorld(unsigned int mx, unsigned int my, double &wx, double &wy) const
    {
        wx = origin_x + (mx + 0.5) * resolution;
        wy = origin_y + (my + 0.5) * resolution;
    }
    bool worldToMap(double wx, double wy, unsigned int &mx, unsigned int &my) const
    {
        if (wx < origin_x || wy < origin_y) return false;
        mx = static_cast<unsigned int>((wx - origin_x) / resolution);
        my = static_cast<unsigned int>((wy - origin_y) / resolution);
        return mx < size_x && my < size_y;
    }
};

// DEP/src/Helpers.cpp:185-216
std::vector<unsigned int> nhood4(unsigned int idx, const Map &costmap)
{
    std::vector<unsigned int> out;
    unsigned int size_x_ = costmap.size_x, size_y_ = costmap.size_y;
    if (idx > size_x_ * size_y_ - 1) return out;
    if (idx % size_x_ > 0) out.push_back(idx - 1);
    if (idx % size_x_ < size_x_ - 1) out.push_back(idx + 1);
    if (idx >= size_x_) out.push_back(idx - size_x_);
    if (idx < size_x_ * (size_y_ - 1)) out.push_back(idx + size_x_);
    return out;
}

// DEP/src/Helpers.cpp:224-255
std::vector<unsigned int> nhood8(unsigned int idx, const Map &costmap)
{
    std::vector<unsigned int> out = nhood4(idx, costmap);
    unsigned int size_x_ = costmap.size_x, size_y_ = costmap.size_y;
    if (idx > size_x_ * size_y_ - 1) return out;
    if (idx % size_x_ > 0 && idx >= size_x_) out.push_back(idx - 1 - size_x_);
    if (idx % size_x_ > 0 && idx < size_x_ * (size_y_ - 1)) out.push_back(idx - 1 + size_x_);
    if (idx % size_x_ < size_x_ - 1 && idx >= size_x_) out.push_back(idx + 1 - size_x_);
    if (idx % size_x_ < size_x_ - 1 && idx < size_x_ * (size_y_ - 1)) out.push_back(idx + 1 + size_x_);
    return out;
}

// DEP/src/Helpers.cpp:285-329 (the `unsigned char val` parameter truncates lethal_threshold like the reference)
bool nearestFreeCell(unsigned int &result, unsigned int start, unsigned char val, const Map &costmap)
{
    const unsigned char *map = costmap.map;
    const unsigned int size_x = costmap.size_x, size_y = costmap.size_y;
    if (start >= size_x * size_y) return false;
    std::queue<unsigned int> bfs;
    std::vector<bool> visited_flag(size_x * size_y, false);
    bfs.push(start);
    visited_flag[start] = true;
    while (!bfs.empty()) {
        unsigned int idx = bfs.front();
        bfs.pop();
        if (map[idx] < val) { result = idx; return true; }
        for (unsigned nbr : nhood8(idx, costmap)) {
            if (!visited_flag[nbr]) { bfs.push(nbr); visited_flag[nbr] = true; }
        }
    }
    return false;
}

struct Search {
    Map costmap_;
    int lethal_threshold_, min_frontier_cluster_size_, max_frontier_cluster_size_;
    double max_frontier_distance_;
    // outputs
    int32_t *cell_piece;          // [ny*nx] sequence number of the piece (emitted or dropped) a cell was collected into, -1
    int32_t *cell_seed;           // [ny*nx] initial_cell of the buildNewFrontier call that claimed the cell, -1
    std::vector<std::pair<double, double>> goals;   // emitted Frontier records, in searchFrom's output order
    std::vector<int> sizes;
    std::vector<int> piece_of_output;               // piece sequence number of each emitted record
    long long n_every = 0;        // every_frontier_list.size()
    int piece_seq = 0;

    // FrontierSearch.hpp:129-142
    bool isLethal(unsigned char value) const { return (int)value >= lethal_threshold_ && value != 255; }
    bool isUnknown(unsigned char value) const { return value == 255; }
    bool isFree(unsigned char value) const { return (int)value < lethal_threshold_; }

    // FrontierSearch.cpp:218-249
    bool isNewFrontierCell(unsigned int idx, const std::vector<bool> &frontier_flag) const
    {
        if (!isUnknown(costmap_.map[idx]) || frontier_flag[idx]) return false;
        bool has_one_free_neighbour = false, has_one_lethal_neighbour = false;
        for (unsigned int nbr : nhood4(idx, costmap_)) {
            if (isFree(costmap_.map[nbr])) has_one_free_neighbour = true;
            if (isLethal(costmap_.map[nbr])) has_one_lethal_neighbour = true;
        }
        if (has_one_lethal_neighbour) return false;
        else if (has_one_free_neighbour) return true;
        else return false;
    }

    // FrontierSearch.hpp:84-127
    std::pair<double, double> getCentroidOfCells(std::vector<std::pair<double, double>> &cells, double distance_to_offset) const
    {
        double sumX = 0, sumY = 0;
        for (const auto &point : cells) { sumX += point.first; sumY += point.second; }
        double centerX = static_cast<double>(sumX) / cells.size();
        double centerY = static_cast<double>(sumY) / cells.size();
        bool offset_centroid = false;
        double varX = 0, varY = 0;
        for (const auto &point : cells) {
            if (sqrt(pow(point.first - centerX, 2) + pow(point.second - centerY, 2)) < costmap_.resolution * 3) offset_centroid = true;
            varX += std::abs(point.first - centerX);
            varY += std::abs(point.second - centerY);
        }
        if (varX > varY && offset_centroid) centerY -= distance_to_offset;
        if (varX < varY && offset_centroid) centerX -= distance_to_offset;
        return std::make_pair(centerX, centerY);
    }

    // FrontierSearch.hpp:156-181
    struct SortByMedianFunctor {
        std::pair<double, double> centroid;
        bool operator()(const std::pair<double, double> &a, const std::pair<double, double> &b) const
        {
            auto angle_a = atan2(a.second - centroid.second, a.first - centroid.first);
            if (angle_a < 0) angle_a = angle_a + (2 * M_PI);
            auto angle_b = atan2(b.second - centroid.second, b.first - centroid.first);
            if (angle_b < 0) angle_b = angle_b + (2 * M_PI);
            if (0 <= angle_a && angle_a <= M_PI / 2 && 3 * M_PI / 2 <= angle_b && angle_b <= 2 * M_PI) return false;
            if (0 <= angle_b && angle_b <= M_PI / 2 && 3 * M_PI / 2 <= angle_a && angle_a <= 2 * M_PI) return true;
            return angle_a < angle_b;
        }
    };

    struct Piece { double gx, gy; int size; int seq; };

    Piece finish_piece(std::vector<std::pair<double, double>> &frontier_cell_indices, int currentFrontierSize)
    {
        // FrontierSearch.cpp:158-170 / 193-205 (FRONTIER_POINT_MEDIAN)
        auto cluster_centroid = getCentroidOfCells(frontier_cell_indices, (costmap_.resolution * 1.414 * 2));
        SortByMedianFunctor sortFunctor{cluster_centroid};
        std::sort(frontier_cell_indices.begin(), frontier_cell_indices.end(), sortFunctor);
        auto goal_point = frontier_cell_indices[static_cast<int>(frontier_cell_indices.size() / 2)];
        frontier_cell_indices.clear();
        return Piece{goal_point.first, goal_point.second, currentFrontierSize, piece_seq};
    }

    // FrontierSearch.cpp:98-216
    std::vector<Piece> buildNewFrontier(unsigned int initial_cell, std::vector<bool> &frontier_flag)
    {
        int currentFrontierSize = 1;
        std::vector<Piece> calculated_frontiers;
        std::vector<std::pair<double, double>> frontier_cell_indices;
        unsigned int ix, iy;
        costmap_.indexToCells(initial_cell, ix, iy);
        double wix, wiy;
        costmap_.mapToWorld(ix, iy, wix, wiy);
        ++n_every;                                                   // every_frontier_list.push_back({wix, wiy})
        frontier_cell_indices.push_back(std::make_pair(wix, wiy));
        cell_piece[initial_cell] = piece_seq;
        cell_seed[initial_cell] = (int32_t)initial_cell;
        std::queue<unsigned int> bfs;
        bfs.push(initial_cell);
        while (!bfs.empty()) {
            unsigned int idx = bfs.front();
            bfs.pop();
            for (unsigned int nbr : nhood8(idx, costmap_)) {
                if (isNewFrontierCell(nbr, frontier_flag)) {
                    frontier_flag[nbr] = true;
                    unsigned int mx, my;
                    double wx, wy;
                    costmap_.indexToCells(nbr, mx, my);
                    costmap_.mapToWorld(mx, my, wx, wy);
                    ++n_every;
                    frontier_cell_indices.push_back(std::make_pair(wx, wy));
                    cell_piece[nbr] = piece_seq;
                    cell_seed[nbr] = (int32_t)initial_cell;
                    currentFrontierSize = currentFrontierSize + 1;
                    bfs.push(nbr);
                    if (currentFrontierSize > max_frontier_cluster_size_) {
                        calculated_frontiers.push_back(finish_piece(frontier_cell_indices, currentFrontierSize));
                        ++piece_seq;
                        currentFrontierSize = 0;
                    }
                }
            }
        }
        if (currentFrontierSize > min_frontier_cluster_size_)
            calculated_frontiers.push_back(finish_piece(frontier_cell_indices, currentFrontierSize));
        ++piece_seq;                                                 // the remainder (emitted or dropped) ends here
        return calculated_frontiers;
    }

    // FrontierSearch.cpp:21-96; returns false when the robot is off the map
    bool searchFrom(double px, double py)
    {
        unsigned int mx, my;
        if (!costmap_.worldToMap(px, py, mx, my)) return false;
        const unsigned int n = costmap_.size_x * costmap_.size_y;
        std::vector<bool> frontier_flag(n, false), visited_flag(n, false);
        std::queue<unsigned int> bfs;
        unsigned int clear, pos = costmap_.getIndex(mx, my);
        if (nearestFreeCell(clear, pos, (unsigned char)lethal_threshold_, costmap_)) bfs.push(clear);
        else bfs.push(pos);
        visited_flag[bfs.front()] = true;
        while (!bfs.empty()) {
            unsigned int idx = bfs.front();
            bfs.pop();
            for (unsigned nbr : nhood4(idx, costmap_)) {
                if (costmap_.map[nbr] < 254 && !visited_flag[nbr]) {
                    visited_flag[nbr] = true;
                    unsigned int nbr_mx, nbr_my;
                    double nbr_wx, nbr_wy;
                    costmap_.indexToCells(nbr, nbr_mx, nbr_my);
                    costmap_.mapToWorld(nbr_mx, nbr_my, nbr_wx, nbr_wy);
                    if (sqrt(pow(px - nbr_wx, 2) + pow(py - nbr_wy, 2)) <
                        max_frontier_distance_ + (max_frontier_cluster_size_ * costmap_.resolution * 1.414))
                        bfs.push(nbr);
                } else if (isNewFrontierCell(nbr, frontier_flag)) {
                    frontier_flag[nbr] = true;
                    for (const Piece &f : buildNewFrontier(nbr, frontier_flag)) {
                        if (f.size > min_frontier_cluster_size_) {
                            goals.push_back(std::make_pair(f.gx, f.gy));
                            sizes.push_back(f.size);
                            piece_of_output.push_back(f.seq);
                        }
                    }
                }
            }
        }
        return true;
    }
};

}  // namespace

extern "C" {

/* Runs FrontierSearch::searchFrom on a 2-D costmap.  cell_piece / cell_seed [ny*nx] (filled with -1 first); goal_xy
 * [max_out][2], size [max_out], piece [max_out]: the Frontier records in output order (only the first max_out are stored);
 * *n_out their number; *n_every = every_frontier_list.size().  Returns 1, or 0 when the robot is off the map
 * (FrontierSearch.cpp:28-33: empty list). */
int fso_frontier_search(const uint8_t *map, int32_t nx, int32_t ny, double origin_x, double origin_y, double resolution,
                        double px, double py, int32_t lethal_threshold, int32_t min_cluster, int32_t max_cluster,
                        double max_distance, int32_t *cell_piece, int32_t *cell_seed, int32_t max_out, double *goal_xy,
                        int32_t *size, int32_t *piece, int32_t *n_out, int64_t *n_every)
{
    Search s;
    s.costmap_ = Map{map, (unsigned)nx, (unsigned)ny, origin_x, origin_y, resolution};
    s.lethal_threshold_ = lethal_threshold;
    s.min_frontier_cluster_size_ = min_cluster;
    s.max_frontier_cluster_size_ = max_cluster;
    s.max_frontier_distance_ = max_distance;
    s.cell_piece = cell_piece;
    s.cell_seed = cell_seed;
    for (int64_t i = 0; i < (int64_t)nx * ny; ++i) { cell_piece[i] = -1; cell_seed[i] = -1; }
    const bool ok = s.searchFrom(px, py);
    const int32_t n = (int32_t)s.goals.size();
    for (int32_t i = 0; i < n && i < max_out; ++i) {
        goal_xy[2 * i] = s.goals[i].first; goal_xy[2 * i + 1] = s.goals[i].second;
        size[i] = s.sizes[i];
        piece[i] = s.piece_of_output[i];
    }
    *n_out = n;
    *n_every = s.n_every;
    return ok ? 1 : 0;
}

}  // extern "C"
