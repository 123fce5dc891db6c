// Where the cuts between subdomains go when they follow the plasma.
//
// The reference partitions the domain once, at t = 0, by the electrons' initial density (GridDesign::balanced,
// grid/mod.rs:157-206, called from main.rs:103-110) and keeps that partition for the whole run.  A laser that piles the
// plasma up (hole boring) or a beam that crosses the domain at c (colliding beams) then leaves most particles on one or
// two subdomains.  opal_recut_plan is the same prefix-sum partition, applied to the particles where they stand NOW and
// held to what one re-cut can move: a cut stays inside the two subdomains it separates (what changes hands then only
// travels between neighbours, through the link the leavers already use), moves `max_shift` cells at the most, and never
// moves more particles of one species than its edge has room for.  Pure host arithmetic on numbers every subdomain holds
// (the all-reduced per-cell histogram), so every subdomain arrives at the same cuts without talking (sim.cpp: sim_recut).
#include <algorithm>
#include <cmath>
#include <vector>

#include "opal_internal.hpp"

using namespace opal;

extern "C" int opal_recut_plan(int numtasks, const size_t *cuts, const double *cost, const double *most, size_t min_size,
                               size_t max_shift, const double *edge_room, size_t *new_cuts, double *imbalance_before,
                               double *imbalance_after)
{
    if (numtasks < 1 || !cuts || !cost || !new_cuts || min_size == 0) {
        set_error("opal_recut_plan: bad arguments");
        return OPAL_ERR_ARG;
    }
    const int N = numtasks;
    const size_t ncells = cuts[N];
    for (int r = 0; r < N; r++)
        if (cuts[r + 1] < cuts[r] + min_size || cuts[0] != 0) {
            set_error("opal_recut_plan: cuts must start at 0 and leave every subdomain %zu cells", min_size);
            return OPAL_ERR_ARG;
        }
    std::vector<double> P(ncells + 1, 0.0), M(ncells + 1, 0.0);  // prefix sums of the cost and of the fullest species' count
    for (size_t c = 0; c < ncells; c++) {
        P[c + 1] = P[c] + cost[c];
        M[c + 1] = M[c] + (most ? most[c] : cost[c]);
    }
    const double total = P[ncells], mean = total / (double)N;
    auto imbalance = [&](const size_t *k) {
        double worst = 0.0;
        for (int r = 0; r < N; r++) worst = std::max(worst, P[k[r + 1]] - P[k[r]]);
        return mean > 0.0 ? worst / mean : 1.0;
    };
    for (int r = 0; r <= N; r++) new_cuts[r] = cuts[r];
    if (imbalance_before) *imbalance_before = imbalance(cuts);
    if (imbalance_after) *imbalance_after = imbalance(cuts);
    if (N == 1 || !(total > 0.0)) return OPAL_OK;
    for (int i = 1; i < N; i++) {
        // the reference's rule (grid/mod.rs:170-176): the first cell at which the running sum reaches i / N of the total ...
        const double target = mean * (double)i;
        size_t k = (size_t)(std::lower_bound(P.begin(), P.end(), target) - P.begin());
        // ... or the cell before it, whichever comes closer (the reference cuts once; a cut that is revisited every few
        // dozen steps should not be biased to one side)
        if (k > 0 && std::fabs(P[k - 1] - target) <= std::fabs(P[k] - target)) k--;
        // held to what one re-cut can move
        const size_t old = cuts[i];
        size_t lo = cuts[i - 1] + min_size, hi = cuts[i + 1] - min_size;  // inside the two subdomains it separates
        if (old > max_shift && old - max_shift > lo) lo = old - max_shift;
        if (old + max_shift < hi) hi = old + max_shift;
        k = std::min(std::max(k, lo), hi);
        const double room = edge_room ? edge_room[i - 1] : -1.0;
        if (room >= 0.0) {
            // no more particles of one species than the edge's messages hold: the cells that change hands and the cell either
            // side (when the hand-over is a step's own push, a particle moves up to a cell meanwhile: sim.cpp, sim_recut C)
            auto at_stake = [&](size_t a, size_t b) { return M[std::min(b + 1, ncells)] - M[a > 0 ? a - 1 : 0]; };  // a < b
            while (k > old && at_stake(old, k) > room) k--;
            while (k < old && at_stake(k, old) > room) k++;
        }
        new_cuts[i] = k;
    }
    // every subdomain keeps min_size cells: a cut that came too close to a neighbour goes back towards where it was (a cut
    // stays min_size cells clear of its neighbours' OLD places, so "back" is always enough, and never past its old place)
    for (int i = 1; i < N; i++)
        if (new_cuts[i] < new_cuts[i - 1] + min_size) new_cuts[i] = std::min(new_cuts[i - 1] + min_size, std::max(cuts[i], new_cuts[i]));
    for (int i = N - 1; i >= 1; i--)
        if (new_cuts[i] + min_size > new_cuts[i + 1]) new_cuts[i] = std::max(new_cuts[i + 1] - min_size, std::min(cuts[i], new_cuts[i]));
    for (int r = 0; r < N; r++)
        if (new_cuts[r + 1] < new_cuts[r] + min_size) {  // (cannot happen, by the argument above: never hand out a bad partition)
            for (int q = 0; q <= N; q++) new_cuts[q] = cuts[q];
            break;
        }
    if (imbalance_after) *imbalance_after = imbalance(new_cuts);
    return OPAL_OK;
}
