// ris_driver.cpp — C++ restatement of the RIS project's host side, talking to the GPU only through the C-ABI.
//
//   Main            BVH_RIS_DL/Assets/Scripts/Main.cs
//   ShaderHelper    BVH_RIS_DL/Assets/Scripts/ShaderHelper.cs
//   MeshHelper      BVH_RIS_DL/Assets/Scripts/MeshHelper.cs (only the layout of what CreateSceneObjects returns)
// The RIS project's scene arrays are derived from a built rth_scene: its BVH builder is the same routine as the
// ReSTIR project's (MeshHelper.cs:196-257 / ObjectHelper.cs:219-290), so nodes, scene objects and lights carry
// over unchanged; triangles are expanded to the 76-byte inline-vertex form and materials narrowed to Material2.
#include "rth_api.h"
#include "unity_math.h"
#include "unity_shim.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

using namespace shim;

struct rth_ris_main_s {
    rt_ctx ctx = nullptr;
    rth_ris_config cfg;
    rth_scene sceneHandle = nullptr;
    rth_scene_data scene;
    bool DoUpdateSettings = false;   // Main.cs:48
    std::vector<rt_ris_tri> Tris;
    std::vector<rt_ris_material> Material2s;
    std::string err;

    ComputeShader rtShader, ppShader, pcShader;
    int FrameCount = 0;
    int framesStarted = 0;
    const int RayTracerThreadSize = 8, PostProcesserThreadSize = 8, PreCalcThreadSize = 256;  // Main.cs:55-57

    ComputeBuffer BVBuffer, TriBuffer, SceneObjectDataBuffer, LightObjectBuffer, MaterialBuffer, CandidateBuffer, HitInfoBuffer;
    RenderTexture RTResultTexture, AccumulatedResultTexture, DebugOverlayTexture, RayHitPointTexture, testTexture;

    int2 Resolution() const { return {cfg.Resolution[0], cfg.Resolution[1]}; }

    // MeshHelper.LoadMesh (:33-73) + Utils.TrisFromTri2s + the parent-index pass (:441-450), from the built scene.
    void BuildRisArrays() {
        Tris.resize((size_t)scene.nTriangles);
        // (on the host's threads: the RIS project reloads with every camera move, so this runs once per frame then)
#pragma omp parallel for schedule(static) if (scene.nTriangles > 16384)
        for (int64_t i = 0; i < scene.nTriangles; i++) {
            const rt_triangle& t = scene.Triangles[i];
            const rt_vertex &a = scene.Vertices[t.vertex0Index], &b = scene.Vertices[t.vertex1Index], &c = scene.Vertices[t.vertex2Index];
            rt_ris_tri r;
            memset(&r, 0, sizeof r);
            r.vA = a.pos; r.vB = b.pos; r.vC = c.pos;
            r.uvA = a.uv; r.uvB = b.uv; r.uvC = c.uv;
            Tris[(size_t)i] = r;
        }
        // every triangle of an object's mesh gets that object's index; with a shared mesh the last object wins
        for (int64_t i = 0; i < scene.nSceneObjects; i++) {
            const rt_int2 range = scene.ComponentDatas[scene.SceneObjects[i].bvStartIndex];
            for (int k = range.x; k < range.x + range.y; k++) Tris[(size_t)k].parentIndex = (int32_t)i;
        }
        // Main.SetData (:271-282): colour + emission strength from MatTypesInput1, smoothness from MatTypesInput2.x
        Material2s.resize((size_t)(scene.nMaterials > 0 ? scene.nMaterials : 1));
        for (auto& m : Material2s) memset(&m, 0, sizeof m);
        for (int64_t i = 0; i < scene.nMaterials; i++) {
            rt_ris_material& m = Material2s[(size_t)i];
            m.color = scene.Materials[i].col;
            m.specularColor = {1, 1, 1};
            m.brightness = scene.Materials[i].brightness;
            m.smoothness = cfg.smoothness;
        }
    }

    // ShaderHelper.cs:7-52
    void SetMaterialBuffer() { rtShader.SetBuffer(0, "Materials", MaterialBuffer); rtShader.SetBuffer(1, "Materials", MaterialBuffer); }
    void SetTriBuffer() {
        rtShader.SetBuffer(0, "Tris", TriBuffer);
        rtShader.SetBuffer(1, "Tris", TriBuffer);
        pcShader.SetBuffer(0, "Tris", TriBuffer);
        pcShader.SetInt("TrisNum", (int)Tris.size());
        rtShader.SetInt("TrisNum", (int)Tris.size());
    }
    void SetBVBuffer() {
        rtShader.SetBuffer(0, "BVs", BVBuffer);
        rtShader.SetBuffer(1, "BVs", BVBuffer);
        rtShader.SetInt("BVsNum", (int)scene.nBVs);
    }
    void SetSceneObjectDataBuffer() {
        rtShader.SetBuffer(0, "SceneObjects", SceneObjectDataBuffer);
        rtShader.SetBuffer(1, "SceneObjects", SceneObjectDataBuffer);
        pcShader.SetBuffer(0, "SceneObjects", SceneObjectDataBuffer);
        rtShader.SetInt("SceneObjectsNum", (int)scene.nSceneObjects);
    }
    void SetLightObjectBuffer() { rtShader.SetBuffer(0, "LightObjects", LightObjectBuffer); }
    void SetCandidateBuffer() { rtShader.SetBuffer(0, "Candidates", CandidateBuffer); rtShader.SetBuffer(1, "Candidates", CandidateBuffer); }
    void SetHitInfoBuffer() { rtShader.SetBuffer(0, "HitInfos", HitInfoBuffer); rtShader.SetBuffer(1, "HitInfos", HitInfoBuffer); }

    // Main.cs:105-120 (without the pause controls)
    void Update() {
        UpdatePerFrame();
        if (DoUpdateSettings) { DoUpdateSettings = false; UpdateSettings(); }
    }
    // Main.cs:132-144
    void UpdatePerFrame() {
        int FrameRand = (int)(((uint64_t)cfg.frameRandBase + (uint64_t)cfg.frameRandStep * (uint64_t)framesStarted) % 999999u);
        framesStarted++;
        rtShader.SetInt("FrameRand", FrameRand);
        rtShader.SetInt("FrameCount", FrameCount);
        ppShader.SetInt("FrameCount", FrameCount);
        FrameCount++;
        SetCameraOrientationAndTransform();
    }
    // Main.cs:183-214
    void SetCameraOrientationAndTransform() {
        rtShader.SetVector("WorldSpaceCameraPos", cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2]);
        M4 CameraTransform;
        rth_ris_camera_transform(cfg.cameraEulerDeg, CameraTransform.m);
        rtShader.SetMatrix("CameraTransform", CameraTransform);
    }
    // Main.cs:221-265
    void UpdateSettings() {
        FrameCount = 0;
        SetData();
        int resolutionArray[2] = {cfg.Resolution[0], cfg.Resolution[1]};
        rtShader.SetInts("Resolution", resolutionArray, 2);
        rtShader.SetInt("MaxBounceCount", cfg.MaxBounceCount);
        rtShader.SetInt("RaysPerPixel", cfg.RaysPerPixel);
        rtShader.SetFloat("ScatterProbability", cfg.ScatterProbability);
        float viewSpaceDims[2];
        rth_view_space_dims(cfg.fieldOfView, cfg.Resolution[0], cfg.Resolution[1], viewSpaceDims);  // same formula, Main.cs:233-238
        rtShader.SetVector("ViewSpaceDims", viewSpaceDims[0], viewSpaceDims[1]);
        rtShader.SetFloat("DefocusStrength", cfg.DefocusStrength);
        rtShader.SetFloat("FocalPlaneFactor", cfg.FocalPlaneFactor);
        int DebugDataMaxValues[2] = {cfg.DebugMaxTriChecks, cfg.DebugMaxBVChecks};
        rtShader.SetInts("DebugDataMaxValues", DebugDataMaxValues, 2);
        rtShader.SetInt("SceneObjectCandidatesNum", cfg.SceneObjectCandidatesNum);
        rtShader.SetInt("TriCandidatesNum", cfg.TriCandidatesNum);
        rtShader.SetInt("TotCandidatesNum", cfg.SceneObjectCandidatesNum * cfg.TriCandidatesNum);
        rtShader.SetInt("SceneObjectReservoirTestsNum", cfg.SceneObjectReservoirTestsNum);
        rtShader.SetInt("TriReservoirTestsNum", cfg.TriReservoirTestsNum);
        rtShader.SetInt("CandidateReservoirTestsNum", cfg.CandidateReservoirTestsNum);
        int texDims[2] = {1, 1};
        rtShader.SetInts("TexDims", texDims, 2);
        rtShader.SetTexture(1, "TestTexture", testTexture);
    }
    // Main.cs:267-312 (+ MeshHelper.CreateSceneObjects' SetInt/SetFloat calls, MeshHelper.cs:365,395,403,422)
    void SetData() {
        ReleaseBuffers();  // ComputeHelper.Release(AllBuffers()), Main.cs:269
        // meshHelper.CreateSceneObjects() (:287): the arrays as the scene holds them now
        if (rth_scene_get(sceneHandle, &scene) != 0) throw DriverError{"the scene is not built"};
        BuildRisArrays();
        if (!testTexture.h) testTexture = CreateTexture(ctx, {1, 1}, RT_FORMAT_RGBA8_UNORM);
        MaterialBuffer = ComputeHelper::CreateStructuredBuffer(ctx, Material2s.data(), (int64_t)Material2s.size(), sizeof(rt_ris_material));
        SetMaterialBuffer();
        rtShader.SetInt("MaxBVHDepth", scene.maxBVHDepth);
        rtShader.SetFloat("TotArea", scene.totArea);
        rtShader.SetInt("SceneBVHStartIndex", scene.sceneBVHStartIndex);
        rtShader.SetInt("EmittingObjectsNum", scene.emittingObjectsNum);
        BVBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.BVs, scene.nBVs, sizeof(rt_bv));
        SetBVBuffer();
        SceneObjectDataBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.SceneObjects, scene.nSceneObjects, sizeof(rt_scene_object));
        SetSceneObjectDataBuffer();
        TriBuffer = ComputeHelper::CreateStructuredBuffer(ctx, Tris.data(), (int64_t)Tris.size(), sizeof(rt_ris_tri));
        SetTriBuffer();
        RunPreCalcShader();
        LightObjectBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.LightObjects, scene.nLightObjects, sizeof(rt_light_object));
        SetLightObjectBuffer();
        int64_t npx = (int64_t)cfg.Resolution[0] * cfg.Resolution[1];
        int64_t ncand = npx * cfg.RaysPerPixel * cfg.SceneObjectCandidatesNum * cfg.TriCandidatesNum;
        CandidateBuffer = ComputeHelper::CreateStructuredBuffer(ctx, ncand > 0 ? ncand : 1, 16);
        SetCandidateBuffer();
        HitInfoBuffer = ComputeHelper::CreateStructuredBuffer(ctx, npx, sizeof(rt_ris_hit_info));
        SetHitInfoBuffer();
    }
    void RunPreCalcShader() { ComputeHelper::DispatchKernel(pcShader, "CalcTriNormals", (int)Tris.size(), PreCalcThreadSize); }  // :314-317
    // Main.cs:319-371
    void CreateTextures() {
        if (RTResultTexture.h) return;
        const rt_format RT_FORMAT_TARGET = cfg.Unorm8RenderTargets ? RT_FORMAT_RGBA8_UNORM : RT_FORMAT_RGBA32F;  // Main.cs:324-364
        RTResultTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(1, "Result", RTResultTexture);
        ppShader.SetTexture(0, "Result", RTResultTexture);
        AccumulatedResultTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        ppShader.SetTexture(0, "AccumulatedResult", AccumulatedResultTexture);
        DebugOverlayTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "DebugOverlay", DebugOverlayTexture);
        RayHitPointTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "RayHitPoints", RayHitPointTexture);
    }
    void RunStage(int stage) {
        CreateTextures();
        if (stage == 0) ComputeHelper::DispatchKernel(rtShader, "GenerateCandidates", Resolution(), RayTracerThreadSize);
        else if (stage == 1) ComputeHelper::DispatchKernel(rtShader, "TraceRays", Resolution(), RayTracerThreadSize);
        else throw DriverError{"unknown stage"};
    }
    void RunRenderShader() { RunStage(0); RunStage(1); }
    void RunPostProcessingShader() { ComputeHelper::DispatchKernel(ppShader, "AccumulateFrames", Resolution(), PostProcesserThreadSize); }  // :373-376
    // Main.cs:87-103
    void Start() {
        if (cfg.RaysPerPixel != 1) cfg.RaysPerPixel = 1;  // forced when the RIS shader is in use (:92-97)
        UpdatePerFrame();
        UpdateSettings();
        framesStarted = 0;  // FrameRand numbering starts with the first rendered frame
    }
    void ReleaseBuffers() {  // AllBuffers(), Main.cs:389
        ComputeBuffer* bufs[] = {&BVBuffer, &TriBuffer, &SceneObjectDataBuffer, &LightObjectBuffer, &MaterialBuffer, &CandidateBuffer, &HitInfoBuffer};
        for (auto* b : bufs) b->Release();
    }
    void ReleaseAll() {
        ReleaseBuffers();
        RenderTexture* texs[] = {&RTResultTexture, &AccumulatedResultTexture, &DebugOverlayTexture, &RayHitPointTexture, &testTexture};
        for (auto* t : texs) t->Release();
        rt_shader_close(rtShader.h); rt_shader_close(ppShader.h); rt_shader_close(pcShader.h);
        rtShader.h = ppShader.h = pcShader.h = nullptr;
    }
};

namespace {
template <class F>
int guarded(rth_ris_main m, F f) {
    if (!m) return -1;
    try {
        f();
        return 0;
    } catch (const DriverError& e) {
        m->err = e.msg;
        return -1;
    }
}
}  // namespace

extern "C" {

// Main.SetCameraOrientationAndTransform, Main.cs:189-211: the matrix is written row by row in the C#; `out` is
// column-major like every matrix that goes through SetMatrix.
void rth_ris_camera_transform(const float eulerDeg[3], float out[16]) {
    const float Deg2Rad = 0.0174532924f;
    float rx = eulerDeg[0] * Deg2Rad, ry = eulerDeg[1] * Deg2Rad, rz = eulerDeg[2] * Deg2Rad;
    float temp = rx;
    rx = ry;
    ry = -temp;
    rz = -rz;
    float cosX = (float)cos((double)rx), sinX = (float)sin((double)rx);
    float cosY = (float)cos((double)ry), sinY = (float)sin((double)ry);
    float cosZ = (float)cos((double)rz), sinZ = (float)sin((double)rz);
    const float rows[4][4] = {
        {cosY * cosZ, cosY * sinZ, -sinY, 0.0f},
        {sinX * sinY * cosZ - cosX * sinZ, sinX * sinY * sinZ + cosX * cosZ, sinX * cosY, 0.0f},
        {cosX * sinY * cosZ + sinX * sinZ, cosX * sinY * sinZ - sinX * cosZ, cosX * cosY, 0.0f},
        {0.0f, 0.0f, 0.0f, 0.0f}};
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[c * 4 + r] = rows[r][c];
}

void rth_ris_default_config(rth_ris_config* c) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->Resolution[0] = 1440; c->Resolution[1] = 810;
    c->fieldOfView = 70;
    c->MaxBounceCount = 1; c->RaysPerPixel = 1;
    c->ScatterProbability = 0.9f; c->DefocusStrength = 0; c->FocalPlaneFactor = 1;
    c->DebugMaxTriChecks = 10; c->DebugMaxBVChecks = 100;
    c->SceneObjectCandidatesNum = 1; c->TriCandidatesNum = 1;
    c->SceneObjectReservoirTestsNum = 1; c->TriReservoirTestsNum = 1; c->CandidateReservoirTestsNum = 1;
    c->cameraPosition[0] = 10; c->cameraPosition[1] = 3; c->cameraPosition[2] = 7;
    // with Main.cs:192-195's axis swap the Euler x angle ends up as the yaw: this looks along (-0.88, 0, 0.47)
    c->cameraEulerDeg[0] = -62; c->cameraEulerDeg[1] = 0; c->cameraEulerDeg[2] = 0;
    c->smoothness = 0;
    c->frameRandBase = 12345; c->frameRandStep = 7919;
}

rth_ris_main rth_ris_main_create(rt_ctx ctx, const rth_ris_config* cfg, rth_scene scene, int row0, int row1) {
    if (!ctx || !cfg || !scene) return nullptr;
    if (cfg->Resolution[0] < 1 || cfg->Resolution[1] < 1 || cfg->SceneObjectCandidatesNum < 0 || cfg->TriCandidatesNum < 0) return nullptr;
    rth_ris_main m = new rth_ris_main_s;
    m->ctx = ctx;
    m->cfg = *cfg;
    m->sceneHandle = scene;
    if (rth_scene_get(scene, &m->scene) != 0) { delete m; return nullptr; }
    int rc = guarded(m, [&] {
        m->rtShader.ctx = m->ppShader.ctx = m->pcShader.ctx = ctx;
        if (rt_set_row_band(ctx, row0, row1) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "RIS_BVH_RayTracer", &m->rtShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "PostProcesser", &m->ppShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "PreCalc", &m->pcShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        m->Start();
    });
    if (rc != 0) {
        fprintf(stderr, "rth_ris_main_create: %s\n", m->err.c_str());
        m->ReleaseAll();
        delete m;
        return nullptr;
    }
    return m;
}
void rth_ris_main_destroy(rth_ris_main m) {
    if (!m) return;
    m->ReleaseAll();
    delete m;
}
const char* rth_ris_main_last_error(rth_ris_main m) { return m ? m->err.c_str() : "null driver"; }

int rth_ris_main_update_per_frame(rth_ris_main m) { return guarded(m, [&] { m->Update(); }); }
int rth_ris_main_reload(rth_ris_main m) {
    if (!m) return -1;
    m->DoUpdateSettings = true;
    return 0;
}
int rth_ris_main_run_stage(rth_ris_main m, int stage) { return guarded(m, [&] { m->RunStage(stage); }); }
int rth_ris_main_run_post_processing(rth_ris_main m) { return guarded(m, [&] { m->CreateTextures(); m->RunPostProcessingShader(); }); }
int rth_ris_main_render_frame(rth_ris_main m) {
    return guarded(m, [&] {
        m->Update();
        m->RunRenderShader();
        m->RunPostProcessingShader();
    });
}
int rth_ris_main_frame_count(rth_ris_main m) { return m ? m->FrameCount : -1; }
rt_ctx rth_ris_main_ctx(rth_ris_main m) { return m ? m->ctx : nullptr; }
int rth_ris_main_resolution(rth_ris_main m, int out[2]) {
    if (!m || !out) return -1;
    out[0] = m->cfg.Resolution[0]; out[1] = m->cfg.Resolution[1];
    return 0;
}

int rth_ris_main_scene(rth_ris_main m, rth_ris_scene_data* out) {
    if (!m || !out) return -1;
    out->Tris = m->Tris.data(); out->nTris = (int64_t)m->Tris.size();
    out->Materials = m->Material2s.data(); out->nMaterials = (int64_t)m->Material2s.size();
    return 0;
}
rt_buffer rth_ris_main_buffer(rth_ris_main m, const char* name) {
    if (!m || !name) return nullptr;
    std::map<std::string, ComputeBuffer*> t = {{"BVs", &m->BVBuffer}, {"Tris", &m->TriBuffer}, {"SceneObjects", &m->SceneObjectDataBuffer},
        {"LightObjects", &m->LightObjectBuffer}, {"Materials", &m->MaterialBuffer}, {"Candidates", &m->CandidateBuffer},
        {"HitInfos", &m->HitInfoBuffer}};
    auto it = t.find(name);
    return it == t.end() ? nullptr : it->second->h;
}
rt_texture rth_ris_main_texture(rth_ris_main m, const char* name) {
    if (!m || !name) return nullptr;
    if (guarded(m, [&] { m->CreateTextures(); }) != 0) return nullptr;
    std::map<std::string, RenderTexture*> t = {{"Result", &m->RTResultTexture}, {"AccumulatedResult", &m->AccumulatedResultTexture},
        {"DebugOverlay", &m->DebugOverlayTexture}, {"RayHitPoints", &m->RayHitPointTexture}};
    auto it = t.find(name);
    return it == t.end() ? nullptr : it->second->h;
}
rt_shader rth_ris_main_shader(rth_ris_main m, const char* name) {
    if (!m || !name) return nullptr;
    if (!strcmp(name, "rt")) return m->rtShader.h;
    if (!strcmp(name, "pp")) return m->ppShader.h;
    if (!strcmp(name, "pc")) return m->pcShader.h;
    return nullptr;
}

}  // extern "C"
