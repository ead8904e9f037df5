// plp_quickhull_dev.hpp -- hand-over between the host part of quickhull's main loop (plp_quickhull_host.hip) and the
// device-resident part (plp_quickhull_dev.hip: the facet graph in device memory, one persistent kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace plp {

struct QhTailHost {
    // in: the facet graph at the hand-over (facet ids = the host's), the compact outside set (point ids ascending)
    int d = 0;
    double tol = 0.0;
    const double* Xdev = nullptr;                  // [N][d] the session's resident points
    std::vector<double> FN, FO;                    // normals [F][d], offsets [F]
    std::vector<int> FV, NB, NBN, CNT, FAR, PQ;    // vertices [F][d]; neighbours [F][capn] + counts; outside counts; furthest
                                                   // point as COMPACT index; the pending facets in queue order
    std::vector<unsigned char> LIVE, INP;
    std::vector<int> opt, oown;                    // compact outside set: point id, owner facet (-1: none)
    std::vector<double> odist;
    int capn = 0;                                  // neighbour slots per facet
    long long total_outside = 0;
    // out: FN, FO, FV, LIVE of all facets made (in creation order), and
    long long iterations = 0, facets_made = 0;
};

// runs the loop to the end; get_block(user, bytes) returns a device block of at least `bytes` (grow-only)
int qh_tail_run(QhTailHost& H, void* (*get_block)(void*, size_t), void* user, hipStream_t st, char* err, size_t errn);

}  // namespace plp
