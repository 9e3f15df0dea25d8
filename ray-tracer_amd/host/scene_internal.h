// scene_internal.h — what scene_builder.cpp shares with scene_cache.cpp (not part of the C interface).
#pragma once

#include <vector>

#include "rth_api.h"

struct SceneArraysRef {
    std::vector<rt_bv>* BVs;
    std::vector<rt_int2>* componentDatas;
    std::vector<rt_vertex>* vertices;
    std::vector<rt_triangle>* triangles;
    std::vector<rt_scene_object>* sceneObjects;
    std::vector<rt_light_object>* lightObjects;
    int* maxBVHDepth;
    int* emittingObjectsNum;
    int* sceneBVHStartIndex;
    float* totArea;
    bool* built;
};
SceneArraysRef rth_scene_arrays_ref(rth_scene s);
bool rth_scene_make_materials(rth_scene s);
