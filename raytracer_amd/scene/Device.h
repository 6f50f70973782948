// Process-wide GPU context used by the scene classes (BVH, KDTree, Renderer).  The device index
// comes from RT_DEVICE (default 0).  Failing to get a gfx950 device is fatal: there is no CPU path.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "../../include/rt_amd.h"

namespace rt {

inline void check(int status, const char* what)
{
    if (status < 0) {
        fprintf(stderr, "raytracer_amd: %s failed (%d): %s\n", what, status, rt_last_error());
        abort();
    }
}

inline rt_context* device()
{
    static rt_context* ctx = nullptr;
    if (!ctx) {
        const char* env = getenv("RT_DEVICE");
        check(rt_context_create(env ? atoi(env) : 0, &ctx), "rt_context_create");
    }
    return ctx;
}

}  // namespace rt
