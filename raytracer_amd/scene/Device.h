// Process-wide GPU contexts used by the scene classes (BVH, KDTree, Renderer): one per GPU named by
// RT_DEVICES (default: the one RT_DEVICE names, else 0).  Failing to get a gfx950 device is fatal: there is no CPU path.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rt_amd.h"

namespace rt {

inline void check(int status, const char* what)
{
    if (status < 0) {
        fprintf(stderr, "raytracer_amd: %s failed (%d): %s\n", what, status, rt_last_error());
        abort();
    }
}

// RT_DEVICES=0,1,2,...: the GPUs the scene is replicated on and the pixel loop is split over (the reference splits
// it over host threads, main.cpp:689-695); default: the single device RT_DEVICE names (0).
inline std::vector<rt_context*>& devices()
{
    static std::vector<rt_context*> ctxs;
    if (ctxs.empty()) {
        std::vector<int> ids;
        if (const char* list = getenv("RT_DEVICES")) {
            for (const char* p = list; *p;) {
                char* end = nullptr;
                long v = strtol(p, &end, 10);
                if (end == p) break;
                ids.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        if (ids.empty()) {
            const char* env = getenv("RT_DEVICE");
            ids.push_back(env ? atoi(env) : 0);
        }
        for (int id : ids) {
            rt_context* c = nullptr;
            check(rt_context_create(id, &c), "rt_context_create");
            ctxs.push_back(c);
        }
    }
    return ctxs;
}

// the first of them: single-ray and batch queries go here
inline rt_context* device() { return devices()[0]; }

}  // namespace rt
