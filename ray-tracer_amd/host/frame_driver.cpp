// frame_driver.cpp — C++ restatement of the reference's frame driver, talking to the GPU only through the C-ABI.
//
// Class and method names follow the C# they restate so the two can be read side by side:
//   Main            BVH_ReSTIR_DL/Assets/Scripts/Main.cs
//   ShaderHelper    BVH_ReSTIR_DL/Assets/Scripts/ShaderHelper.cs
//   ComputeHelper   BVH_ReSTIR_DL/Assets/Scripts/ComputeHelper.cs
// ComputeShader / ComputeBuffer / RenderTexture below are thin wrappers over include/rt_native.h with Unity's
// method names: the same substitution a C# host would make with a P/Invoke shim.
#include "rth_api.h"
#include "unity_math.h"
#include "unity_shim.h"

#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

using namespace shim;

// ---- Main.cs -----------------------------------------------------------------------------------------------------
struct rth_main_s {
    rt_ctx ctx = nullptr;
    rth_main_config cfg;
    rth_scene sceneHandle = nullptr;   // ObjectHelper: SetData asks it for the scene arrays again on every reload
    rth_scene_data scene;
    std::string err;
    bool DoUpdateSettings = false, DoReloadData = false;   // Main.cs:84-85

    ComputeShader rtShader, ppShader, pcShader;
    int FrameCount = 0;
    int framesStarted = 0;  // UpdatePerFrame calls since Start, selects FrameRand
    const int RayTracerThreadSize = 8, PostProcesserThreadSize = 8, PreCalcThreadSize = 256;  // Main.cs:64-66

    ComputeBuffer BVBuffer, RenderTriangleBuffer, VertexBuffer, SceneObjectDataBuffer, LightObjectBuffer, MaterialBuffer;
    ComputeBuffer CandidateBuffer, CandidateReuseBuffer, TemporalFrameBuffer, HitInfoBuffer;
    RenderTexture RTResultTexture, AccumulatedResultTexture, DebugOverlayTexture, RayHitPointATexture, RayHitPointBTexture,
        DepthBufferTexture, NormalsBufferTexture, TextureAtlas, EnvironmentMapTexture;
    bool texturesCreated = false;
    bool haveLastCamera = false;
    float lastWorldSpaceCameraPos[3];
    M4 lastCameraTransform;

    int2 Resolution() const { return {cfg.Resolution[0], cfg.Resolution[1]}; }

    // ShaderHelper.cs:7-73
    void SetMaterialBuffer() { rtShader.SetBuffer(0, "Materials", MaterialBuffer); rtShader.SetBuffer(4, "Materials", MaterialBuffer); }
    void SetTriBuffer() {
        rtShader.SetBuffer(0, "Triangles", RenderTriangleBuffer);
        rtShader.SetBuffer(4, "Triangles", RenderTriangleBuffer);
        pcShader.SetBuffer(0, "Triangles", RenderTriangleBuffer);
        pcShader.SetInt("TrianglesNum", (int)scene.nTriangles);
        rtShader.SetInt("TrianglesNum", (int)scene.nTriangles);
    }
    void SetVertexBuffer() {
        rtShader.SetBuffer(0, "Vertices", VertexBuffer);
        rtShader.SetBuffer(4, "Vertices", VertexBuffer);
        pcShader.SetBuffer(0, "Vertices", VertexBuffer);
    }
    void SetBVBuffer() {
        rtShader.SetBuffer(0, "BVs", BVBuffer);
        rtShader.SetBuffer(4, "BVs", BVBuffer);
        rtShader.SetInt("BVsNum", (int)scene.nBVs);
    }
    void SetSceneObjectDataBuffer() {
        rtShader.SetBuffer(0, "SceneObjects", SceneObjectDataBuffer);
        rtShader.SetBuffer(4, "SceneObjects", SceneObjectDataBuffer);
        pcShader.SetBuffer(0, "SceneObjects", SceneObjectDataBuffer);
        rtShader.SetInt("SceneObjectsNum", (int)scene.nSceneObjects);
    }
    void SetLightObjectBuffer() { rtShader.SetBuffer(0, "LightObjects", LightObjectBuffer); }
    void SetCandidateBuffer() { for (int k = 0; k < 5; k++) rtShader.SetBuffer(k, "Candidates", CandidateBuffer); }
    void SetCandidateReuseBuffer() { rtShader.SetBuffer(1, "CandidatesB", CandidateReuseBuffer); rtShader.SetBuffer(2, "CandidatesB", CandidateReuseBuffer); }
    void SetTemporalFrameBuffer() { rtShader.SetBuffer(3, "TemporalFrameBuffer", TemporalFrameBuffer); rtShader.SetBuffer(4, "TemporalFrameBuffer", TemporalFrameBuffer); }
    void SetHitInfoBuffer() { rtShader.SetBuffer(0, "HitInfos", HitInfoBuffer); rtShader.SetBuffer(4, "HitInfos", HitInfoBuffer); }

    // Main.cs:117-132 (without the pause controls): the per-frame constants, then - when an inspector value changed
    // or a scene object moved since the last frame (OnValidate :206-209, SceneObjectSettings.cs:18-36) - every setting
    // again and, with them, the scene data.  UpdateSettings runs AFTER UpdatePerFrame: the frame that reloads is
    // rendered with the FrameCount already pushed, on freshly zeroed per-pixel buffers, and the count restarts at 0
    // with the frame after it (:213).
    void Update() {
        UpdatePerFrame();
        if (DoUpdateSettings) { DoUpdateSettings = false; UpdateSettings(DoReloadData); DoReloadData = false; }
    }
    // Main.cs:134-145
    void UpdatePerFrame() {
        int FrameRand = (int)(((uint64_t)cfg.frameRandBase + (uint64_t)cfg.frameRandStep * (uint64_t)framesStarted) % 999999u);
        framesStarted++;
        rtShader.SetInt("FrameRand", FrameRand);
        rtShader.SetInt("FrameCount", FrameCount);
        ppShader.SetInt("FrameCount", FrameCount);
        FrameCount++;
        SetCameraData();
    }
    // Main.cs:184-204
    void SetCameraData() {
        const float* worldSpaceCameraPos = cfg.cameraPosition;
        if (!haveLastCamera) memcpy(lastWorldSpaceCameraPos, worldSpaceCameraPos, 12);
        rtShader.SetVector("WorldSpaceCameraPos", worldSpaceCameraPos[0], worldSpaceCameraPos[1], worldSpaceCameraPos[2]);
        rtShader.SetVector("LastWorldSpaceCameraPos", lastWorldSpaceCameraPos[0], lastWorldSpaceCameraPos[1], lastWorldSpaceCameraPos[2]);
        M4 cameraTransform, unusedInverse;
        rth_camera_matrices(cfg.cameraPosition, cfg.cameraRotation, cameraTransform.m, unusedInverse.m);
        if (!haveLastCamera) lastCameraTransform = cameraTransform;
        rtShader.SetMatrix("CameraTransform", cameraTransform);
        rtShader.SetMatrix("LastCameraTransformInverse", inverse(lastCameraTransform));
        memcpy(lastWorldSpaceCameraPos, worldSpaceCameraPos, 12);
        lastCameraTransform = cameraTransform;
        haveLastCamera = true;
    }
    // Main.cs:211-271
    void UpdateSettings(bool resetBufferData) {
        FrameCount = 0;
        if (resetBufferData) SetData();
        int resolutionArray[2] = {cfg.Resolution[0], cfg.Resolution[1]};
        rtShader.SetInts("Resolution", resolutionArray, 2);
        rtShader.SetInt("MaxBounceCount", cfg.MaxBounceCount);
        rtShader.SetFloat("ScatterProbability", cfg.ScatterProbability);
        float viewSpaceDims[2];
        rth_view_space_dims(cfg.fieldOfView, cfg.Resolution[0], cfg.Resolution[1], viewSpaceDims);
        rtShader.SetVector("ViewSpaceDims", viewSpaceDims[0], viewSpaceDims[1]);
        rtShader.SetFloat("DefocusStrength", cfg.DefocusStrength);
        rtShader.SetFloat("FocalPlaneFactor", cfg.FocalPlaneFactor);
        int DebugDataMaxValues[2] = {cfg.DebugMaxTriChecks, cfg.DebugMaxBVChecks};
        rtShader.SetInts("DebugDataMaxValues", DebugDataMaxValues, 2);
        rtShader.SetInt("SceneObjectReservoirTestsNum", cfg.SceneObjectReservoirTestsNum);
        rtShader.SetInt("TriReservoirTestsNum", cfg.TriReservoirTestsNum);
        rtShader.SetInt("CandidateReservoirTestsNum", cfg.CandidateReservoirTestsNum);
        rtShader.SetFloat("TemporalReuseWeight", cfg.TemporalReuseWeight);
        rtShader.SetFloat("PixelMovementThreshold", cfg.PixelMovementThreshold);
        rtShader.SetFloat("SpatialHitPointDiffThreshold", cfg.SpatialHitPointDiffThreshold);
        rtShader.SetFloat("SpatialNormalsAngleThreshold", cfg.SpatialNormalsAngleThreshold);
        rtShader.SetFloat("SpatialBRDFThreshold", cfg.SpatialBRDFThreshold);
        rtShader.SetInt("TemporalCandidatesNum", cfg.TemporalCandidatesNum);
        rtShader.SetFloat("TemporalPrecisionThreshold", cfg.TemporalPrecisionThreshold);
        rtShader.SetFloat("VisibilityReuseThreshold", cfg.VisibilityReuseThreshold);
        if (cfg.DoVisibilityReuse) rtShader.EnableKeyword("VIS_REUSE"); else rtShader.DisableKeyword("VIS_REUSE");
        if (cfg.DoWeightRecalc) rtShader.EnableKeyword("WEIGHT_RECALC"); else rtShader.DisableKeyword("WEIGHT_RECALC");
        if (cfg.DoBRDF) rtShader.EnableKeyword("BRDF"); else rtShader.DisableKeyword("BRDF");
        // Object textures (Main.cs:259-267): the atlas ConstructTextureAtlas packed (1x1 black when no material has a
        // texture); the environment map is a 1x1 sky colour (the reference's HDR map is not in the snapshot).
        int textureAtlasDims[2] = {scene.atlasWidth, scene.atlasHeight};
        rtShader.SetInts("TextureAtlasDims", textureAtlasDims, 2);
        rtShader.SetTexture(0, "TextureAtlas", TextureAtlas);
        rtShader.SetTexture(4, "TextureAtlas", TextureAtlas);
        int environmentMapTexDims[2] = {1, 1};
        rtShader.SetInts("EnvironmentMapTexDims", environmentMapTexDims, 2);
        rtShader.SetTexture(4, "EnvironmentMap", EnvironmentMapTexture);
    }
    // Main.cs:273-310 (+ ObjectHelper.ConstructScene's SetInt/SetFloat calls, ObjectHelper.cs:637,647,686,696)
    void SetData() {
        ReleaseBuffers();  // ComputeHelper.Release(AllBuffers()), Main.cs:275
        // objectHelper.ConstructScene() (:278): the arrays as the scene holds them now
        if (rth_scene_get(sceneHandle, &scene) != 0) throw DriverError{"the scene is not built"};
        rtShader.SetInt("MaxBVHDepth", scene.maxBVHDepth);
        rtShader.SetFloat("TotArea", scene.totArea);
        rtShader.SetInt("SceneBVHStartIndex", scene.sceneBVHStartIndex);
        rtShader.SetInt("EmittingObjectsNum", scene.emittingObjectsNum);
        if (!TextureAtlas.h) {
            TextureAtlas = CreateTexture(ctx, {scene.atlasWidth, scene.atlasHeight}, RT_FORMAT_RGBA8_UNORM);
            int rca = rt_texture_set_data(TextureAtlas.h, scene.atlas, (size_t)scene.atlasWidth * scene.atlasHeight * 4);
            if (rca < 0) throw DriverError{std::string("TextureAtlas upload: ") + rt_last_error(ctx)};
            EnvironmentMapTexture = CreateTexture(ctx, {1, 1}, RT_FORMAT_RGBA32F);
            const float sky[4] = {0.5f, 0.7f, 1.0f, 1.0f};
            int rc = rt_texture_set_data(EnvironmentMapTexture.h, sky, sizeof sky);
            if (rc < 0) throw DriverError{std::string("EnvironmentMap upload: ") + rt_last_error(ctx)};
        }
        MaterialBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.Materials, scene.nMaterials, sizeof(rt_material));
        SetMaterialBuffer();
        BVBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.BVs, scene.nBVs, sizeof(rt_bv));
        SetBVBuffer();
        SceneObjectDataBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.SceneObjects, scene.nSceneObjects, sizeof(rt_scene_object));
        SetSceneObjectDataBuffer();
        RenderTriangleBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.Triangles, scene.nTriangles, sizeof(rt_triangle));
        SetTriBuffer();
        VertexBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.Vertices, scene.nVertices, sizeof(rt_vertex));
        SetVertexBuffer();
        RunPreCalcShader();
        LightObjectBuffer = ComputeHelper::CreateStructuredBuffer(ctx, scene.LightObjects, scene.nLightObjects, sizeof(rt_light_object));
        SetLightObjectBuffer();
        int64_t npx = (int64_t)cfg.Resolution[0] * cfg.Resolution[1];
        CandidateBuffer = ComputeHelper::CreateStructuredBuffer(ctx, npx, sizeof(rt_reservoir));
        SetCandidateBuffer();
        CandidateReuseBuffer = ComputeHelper::CreateStructuredBuffer(ctx, npx, sizeof(rt_reservoir));
        SetCandidateReuseBuffer();
        TemporalFrameBuffer = ComputeHelper::CreateStructuredBuffer(ctx, npx, sizeof(rt_reservoir));
        SetTemporalFrameBuffer();
        HitInfoBuffer = ComputeHelper::CreateStructuredBuffer(ctx, npx, sizeof(rt_hit_info));
        SetHitInfoBuffer();
    }
    // Main.cs:312-380
    void CreateTextures() {
        if (texturesCreated) return;
        // TextureHelper.CreateTexture gives every 3- and 4-channel target the default 8-bit format (TextureHelper.cs:144-155)
        const rt_format RT_FORMAT_TARGET = cfg.Unorm8RenderTargets ? RT_FORMAT_RGBA8_UNORM : RT_FORMAT_RGBA32F;
        RTResultTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(4, "Result", RTResultTexture);
        ppShader.SetTexture(0, "Result", RTResultTexture);
        AccumulatedResultTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        ppShader.SetTexture(0, "AccumulatedResult", AccumulatedResultTexture);
        DebugOverlayTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "DebugOverlay", DebugOverlayTexture);
        rtShader.SetTexture(3, "DebugOverlay", DebugOverlayTexture);
        RayHitPointATexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "RayHitPointsA", RayHitPointATexture);
        rtShader.SetTexture(3, "RayHitPointsA", RayHitPointATexture);
        RayHitPointBTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "RayHitPointsB", RayHitPointBTexture);
        rtShader.SetTexture(3, "RayHitPointsB", RayHitPointBTexture);
        DepthBufferTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_R32F);
        for (int k = 0; k < 4; k++) rtShader.SetTexture(k, "DepthBuffer", DepthBufferTexture);
        NormalsBufferTexture = CreateTexture(ctx, Resolution(), RT_FORMAT_TARGET);
        rtShader.SetTexture(0, "NormalsBuffer", NormalsBufferTexture);
        rtShader.SetTexture(1, "NormalsBuffer", NormalsBufferTexture);
        texturesCreated = true;
    }
    void RunPreCalcShader() { ComputeHelper::DispatchKernel(pcShader, "CalcTriNormals", (int)scene.nTriangles, PreCalcThreadSize); }  // :382-385
    // Main.cs:387-402
    void SpatialReuse() {
        bool reuseBufferCycle = false;
        for (int i = 0; i < cfg.SpatialReuseIterations; i++) {
            reuseBufferCycle = !reuseBufferCycle;
            int maxOffset = (int)(float)pow(3.0, (double)i) * cfg.SpatialReuseBlur;  // (int)Mathf.Pow(3, i)
            rtShader.SetInt("MaxOffset", maxOffset);
            rtShader.SetBool("ReuseBufferCycle", reuseBufferCycle);
            ComputeHelper::DispatchKernel(rtShader, "SpatialReusePass", Resolution(), RayTracerThreadSize);
        }
        if (reuseBufferCycle == true) ComputeHelper::DispatchKernel(rtShader, "TransferToOriginal", Resolution(), RayTracerThreadSize);
    }
    // One iteration of the loop in SpatialReuse() (Main.cs:390-399), and the copy-back after it (:401).  Split out so
    // a multi-GPU host can exchange reservoir halos between passes.
    void SpatialReusePass(int i) {
        bool reuseBufferCycle = (i % 2) == 0;
        int maxOffset = (int)(float)pow(3.0, (double)i) * cfg.SpatialReuseBlur;
        rtShader.SetInt("MaxOffset", maxOffset);
        rtShader.SetBool("ReuseBufferCycle", reuseBufferCycle);
        ComputeHelper::DispatchKernel(rtShader, "SpatialReusePass", Resolution(), RayTracerThreadSize);
    }
    void TransferIfNeeded() {
        if (cfg.SpatialReuseIterations % 2 == 1) ComputeHelper::DispatchKernel(rtShader, "TransferToOriginal", Resolution(), RayTracerThreadSize);
    }
    // Main.cs:404-415
    void RunReSTIRShader() {
        CreateTextures();
        ComputeHelper::DispatchKernel(rtShader, "InitialTrace", Resolution(), RayTracerThreadSize);
        if (cfg.TemporalReuseWeight > 0) ComputeHelper::DispatchKernel(rtShader, "TemporalReuse", Resolution(), RayTracerThreadSize);
        if (cfg.SpatialReuseIterations > 0) SpatialReuse();
        ComputeHelper::DispatchKernel(rtShader, "TraceRays", Resolution(), RayTracerThreadSize);
    }
    void RunStage(int stage) {
        CreateTextures();
        switch (stage) {
            case 0: ComputeHelper::DispatchKernel(rtShader, "InitialTrace", Resolution(), RayTracerThreadSize); break;
            case 1: if (cfg.TemporalReuseWeight > 0) ComputeHelper::DispatchKernel(rtShader, "TemporalReuse", Resolution(), RayTracerThreadSize); break;
            case 2: if (cfg.SpatialReuseIterations > 0) SpatialReuse(); break;
            case 3: ComputeHelper::DispatchKernel(rtShader, "TraceRays", Resolution(), RayTracerThreadSize); break;
            default: throw DriverError{"unknown stage"};
        }
    }
    void RunPostProcessingShader() { ComputeHelper::DispatchKernel(ppShader, "AccumulateFrames", Resolution(), PostProcesserThreadSize); }  // :417-420
    // Main.cs:105-115
    void Start() {
        UpdatePerFrame();
        UpdateSettings(true);
        SetCameraData();
        framesStarted = 0;  // FrameRand numbering starts with the first rendered frame
    }
    void ReleaseBuffers() {  // AllBuffers(), Main.cs:507
        ComputeBuffer* bufs[] = {&BVBuffer, &RenderTriangleBuffer, &VertexBuffer, &SceneObjectDataBuffer, &LightObjectBuffer, &MaterialBuffer,
                                 &CandidateBuffer, &CandidateReuseBuffer, &TemporalFrameBuffer, &HitInfoBuffer};
        for (auto* b : bufs) b->Release();
    }
    void ReleaseAll() {  // Main.cs:509-513
        ReleaseBuffers();
        RenderTexture* texs[] = {&RTResultTexture, &AccumulatedResultTexture, &DebugOverlayTexture, &RayHitPointATexture, &RayHitPointBTexture,
                                 &DepthBufferTexture, &NormalsBufferTexture, &TextureAtlas, &EnvironmentMapTexture};
        for (auto* t : texs) t->Release();
        rt_shader_close(rtShader.h); rt_shader_close(ppShader.h); rt_shader_close(pcShader.h);
        rtShader.h = ppShader.h = pcShader.h = nullptr;
    }
};

namespace {
template <class F>
int guarded(rth_main m, F f) {
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

// Main.UpdateSettings, Main.cs:223-228
void rth_view_space_dims(float fieldOfView, int resX, int resY, float out[2]) {
    float aspectRatio = (float)(resX / resY);  // integer division, Main.cs:223
    float fieldOfViewRad = fieldOfView * 3.14159274f / 180;
    float viewSpaceHeight = (float)tan((double)(fieldOfViewRad * 0.5f));  // Mathf.Tan
    float viewSpaceWidth = aspectRatio * viewSpaceHeight;
    out[0] = viewSpaceWidth;
    out[1] = viewSpaceHeight;
}
// transform.localToWorldMatrix and its inverse, Main.cs:194-199
void rth_camera_matrices(const float position[3], const float rotation[4], float outTransform[16], float outInverse[16]) {
    M4 t = fromPositionRotation(position, rotation);
    M4 i = inverse(t);
    memcpy(outTransform, t.m, 64);
    memcpy(outInverse, i.m, 64);
}

void rth_main_default_config(rth_main_config* c) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->Resolution[0] = 1920; c->Resolution[1] = 1080;
    c->fieldOfView = 70;
    c->MaxBounceCount = 1;
    c->ScatterProbability = 1; c->DefocusStrength = 0; c->FocalPlaneFactor = 1;
    c->DebugMaxTriChecks = 10; c->DebugMaxBVChecks = 100;
    c->SceneObjectReservoirTestsNum = 5; c->TriReservoirTestsNum = 5; c->CandidateReservoirTestsNum = 5;
    c->SpatialReuseIterations = 3; c->TemporalCandidatesNum = 9; c->SpatialReuseBlur = 3;
    c->TemporalPrecisionThreshold = 0.000001f; c->TemporalReuseWeight = 0.9f; c->PixelMovementThreshold = 1;
    c->SpatialHitPointDiffThreshold = 0.5f; c->SpatialNormalsAngleThreshold = 360; c->SpatialBRDFThreshold = 999;
    c->VisibilityReuseThreshold = 5;
    c->DoVisibilityReuse = 1; c->DoWeightRecalc = 0; c->DoBRDF = 1;
    c->cameraPosition[0] = 10; c->cameraPosition[1] = 3; c->cameraPosition[2] = 7;
    c->cameraRotation[0] = 0; c->cameraRotation[1] = -0.5150381f; c->cameraRotation[2] = 0; c->cameraRotation[3] = 0.85716736f;
    c->frameRandBase = 12345; c->frameRandStep = 7919;
}

rth_main rth_main_create(rt_ctx ctx, const rth_main_config* cfg, rth_scene scene, int row0, int row1) {
    if (!ctx || !cfg || !scene) return nullptr;
    rth_main m = new rth_main_s;
    m->ctx = ctx;
    m->cfg = *cfg;
    m->sceneHandle = scene;
    if (rth_scene_get(scene, &m->scene) != 0) { delete m; return nullptr; }
    if (cfg->Resolution[0] < 1 || cfg->Resolution[1] < 1) { delete m; return nullptr; }
    int rc = guarded(m, [&] {
        m->rtShader.ctx = m->ppShader.ctx = m->pcShader.ctx = ctx;
        if (rt_set_row_band(ctx, row0, row1) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "ReSTIR_BVH_RayTracer", &m->rtShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "PostProcessor", &m->ppShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        if (rt_shader_open(ctx, "PreCalc", &m->pcShader.h) < 0) throw DriverError{rt_last_error(ctx)};
        m->Start();
    });
    if (rc != 0) {
        fprintf(stderr, "rth_main_create: %s\n", m->err.c_str());
        m->ReleaseAll();
        delete m;
        return nullptr;
    }
    return m;
}
void rth_main_destroy(rth_main m) {
    if (!m) return;
    m->ReleaseAll();
    delete m;
}
const char* rth_main_last_error(rth_main m) { return m ? m->err.c_str() : "null driver"; }

int rth_main_update_per_frame(rth_main m) { return guarded(m, [&] { m->Update(); }); }
int rth_main_reload(rth_main m, const rth_main_config* cfg) {
    if (!m) return -1;
    if (cfg) {
        // the render targets are created once, at the size of the first frame (CreateTextures, Main.cs:312-380)
        if (cfg->Resolution[0] != m->cfg.Resolution[0] || cfg->Resolution[1] != m->cfg.Resolution[1] ||
            cfg->Unorm8RenderTargets != m->cfg.Unorm8RenderTargets) {
            m->err = "Resolution and the render-target format cannot change on a reload";
            return -1;
        }
        m->cfg = *cfg;
    }
    m->DoUpdateSettings = true;
    m->DoReloadData = true;
    return 0;
}
int rth_main_run_restir_shader(rth_main m) { return guarded(m, [&] { m->RunReSTIRShader(); }); }
int rth_main_run_post_processing(rth_main m) { return guarded(m, [&] { m->RunPostProcessingShader(); }); }
int rth_main_run_stage(rth_main m, int stage) { return guarded(m, [&] { m->RunStage(stage); }); }
int rth_main_run_spatial_pass(rth_main m, int i) { return guarded(m, [&] { m->CreateTextures(); m->SpatialReusePass(i); }); }
int rth_main_run_transfer(rth_main m) { return guarded(m, [&] { m->CreateTextures(); m->TransferIfNeeded(); }); }
int rth_main_render_frame(rth_main m) {
    return guarded(m, [&] {
        m->Update();
        m->RunReSTIRShader();
        m->RunPostProcessingShader();
    });
}
int rth_main_frame_count(rth_main m) { return m ? m->FrameCount : -1; }
int rth_main_set_camera(rth_main m, const float position[3], const float rotation[4]) {
    if (!m || !position || !rotation) return -1;
    memcpy(m->cfg.cameraPosition, position, 12);
    memcpy(m->cfg.cameraRotation, rotation, 16);
    return 0;
}

rt_buffer rth_main_buffer(rth_main m, const char* name) {
    if (!m || !name) return nullptr;
    std::map<std::string, ComputeBuffer*> t = {{"BVs", &m->BVBuffer}, {"Triangles", &m->RenderTriangleBuffer}, {"Vertices", &m->VertexBuffer},
        {"SceneObjects", &m->SceneObjectDataBuffer}, {"LightObjects", &m->LightObjectBuffer}, {"Materials", &m->MaterialBuffer},
        {"Candidates", &m->CandidateBuffer}, {"CandidatesB", &m->CandidateReuseBuffer}, {"TemporalFrameBuffer", &m->TemporalFrameBuffer},
        {"HitInfos", &m->HitInfoBuffer}};
    auto it = t.find(name);
    return it == t.end() ? nullptr : it->second->h;
}
rt_texture rth_main_texture(rth_main m, const char* name) {
    if (!m || !name) return nullptr;
    int rc = guarded(m, [&] { m->CreateTextures(); });
    if (rc != 0) return nullptr;
    std::map<std::string, RenderTexture*> t = {{"Result", &m->RTResultTexture}, {"AccumulatedResult", &m->AccumulatedResultTexture},
        {"DebugOverlay", &m->DebugOverlayTexture}, {"DepthBuffer", &m->DepthBufferTexture}, {"NormalsBuffer", &m->NormalsBufferTexture},
        {"RayHitPointsA", &m->RayHitPointATexture}, {"RayHitPointsB", &m->RayHitPointBTexture}, {"TextureAtlas", &m->TextureAtlas},
        {"EnvironmentMap", &m->EnvironmentMapTexture}};
    auto it = t.find(name);
    return it == t.end() ? nullptr : it->second->h;
}
rt_shader rth_main_shader(rth_main m, const char* name) {
    if (!m || !name) return nullptr;
    if (!strcmp(name, "rt")) return m->rtShader.h;
    if (!strcmp(name, "pp")) return m->ppShader.h;
    if (!strcmp(name, "pc")) return m->pcShader.h;
    return nullptr;
}

}  // extern "C"

// ==================================================================================================================
// Multi-GPU frame loop: one process per GPU, the image cut into row bands, halos and the displayed image moved with
// RCCL point-to-point transfers.  The reference is single-GPU; this is the sharding the path admits (every kernel
// reads only its own pixel except SpatialReusePass, which reads reservoirs and depths up to MaxOffset + 1 rows away).
// The schedule is the one ray-tracer_amd/multigpu.py runs over torch.distributed (and that the CPU tests check
// against the single-process result); here no Python runs between frames.
//
// Streams.  Kernels run on the context's dispatch streams (the library alternates consecutive frames between two), all
// transfers on its transfer stream (rt_transfer_stream), with ONE communicator: every rank enqueues the same sequence
// of grouped send/recv calls on that one stream (halo 0, halo 1, gather 0, halo 2, gather 1, ...; with the per-step
// timing on: halo f, gather f, halo f+1, ... - so the timing switch must be the same on every rank), which is the
// ordering RCCL asks for.  Per frame:
//   dispatch stream : InitialTrace, TemporalReuse | pass 0 on the interior rows | pass 0 on the border rows, later
//                     passes, TransferToOriginal | TraceRays, AccumulateFrames
//   transfer stream :                             | halo exchange, then the gather of the frame before ...
// The interior rows of the first pass read nothing a neighbour owns, so they run while the halos travel; TraceRays
// of frame f+1 is the first kernel that writes what gather f reads (this rank's band of Result) and waits for it.
// ==================================================================================================================
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>

namespace {

// The RCCL entry points used, resolved at run time so that the library loads on machines without RCCL.
struct Rccl {
    typedef struct ncclComm* comm_t;
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, void*) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
    Rccl() {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { why = "librccl.so could not be loaded"; return; }
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) why = std::string("RCCL symbol missing: ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        ok = why.empty();
    }
};
Rccl& rccl() { static Rccl r; return r; }
// The few HIP runtime calls the rehearsal transport below makes, resolved at run time like RCCL's (this library is
// plain C++ and otherwise reaches the GPU only through the C-ABI).
struct HipRt {
    int (*SetDevice)(int) = nullptr;
    int (*EventCreateWithFlags)(void**, unsigned) = nullptr;
    int (*EventRecord)(void*, void*) = nullptr;
    int (*EventDestroy)(void*) = nullptr;
    int (*StreamWaitEvent)(void*, void*, unsigned) = nullptr;
    int (*MemcpyAsync)(void*, const void*, size_t, int, void*) = nullptr;
    bool ok = false;
    HipRt() {
        void* h = nullptr;
        for (const char* name : {"libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); all = all && p; return p; };
        SetDevice = (decltype(SetDevice))sym("hipSetDevice");
        EventCreateWithFlags = (decltype(EventCreateWithFlags))sym("hipEventCreateWithFlags");
        EventRecord = (decltype(EventRecord))sym("hipEventRecord");
        EventDestroy = (decltype(EventDestroy))sym("hipEventDestroy");
        StreamWaitEvent = (decltype(StreamWaitEvent))sym("hipStreamWaitEvent");
        MemcpyAsync = (decltype(MemcpyAsync))sym("hipMemcpyAsync");
        ok = all;
    }
};
HipRt& hipRt() { static HipRt r; return r; }
const int NCCL_UINT8 = 1;   // ncclUint8
const int NCCL_DOUBLE = 8;  // ncclFloat64
const int NCCL_SUM = 0;

// stamps (rt_mark slots) of one frame, for the per-rank breakdown
enum { MK_FRAME = 0, MK_PRE, MK_INTERIOR, MK_HALO0, MK_HALO1, MK_POST0, MK_END, MK_GATHER0, MK_GATHER1,
       // ordering points of the frame loop (rt_wait_mark)
       MK_HALO_DONE = 16, MK_FRAME_END, MK_GATHER_DONE };

}  // namespace

// Rehearsal transport: the ranks of a frame loop as THREADS of one process whose contexts share one GPU - the one case
// RCCL refuses - so that rth_mg_render_frames itself (its order of halo exchanges and gathers, its waits) can be run
// and checked on a single GPU.  Same contract as RCCL's grouped point-to-point calls: the k-th send of rank a to rank
// b meets the k-th receive of b from a; a transfer starts when both transfer streams have reached it and both streams
// go on when it is done.  Carried out as a device copy on the receiver's stream between two events; the host threads
// meet at every group (a receive blocks until its send is posted, a group ends when its sends are taken).
struct rth_hub_s {
    struct SendRec { const void* ptr; size_t bytes; void* ready; void* copied = nullptr; bool taken = false; };
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<std::shared_ptr<SendRec>>> queue;   // [src * world + dst]
    bool broken = false;   // a rank gave up: nobody waits any longer
    // The events of finished transfers.  They are destroyed with the hub, not when the sender has enqueued its wait:
    // a wait on them may still be pending on a stream of another rank's thread, and nothing is gained by finding out
    // what the runtime makes of an event destroyed under it.
    std::vector<void*> spent;
};

struct rth_mg_s {
    rth_hub_s* hub = nullptr;    // rehearsal transport instead of a communicator
    rth_main m = nullptr;        // the ReSTIR driver, or
    rth_ris_main ris = nullptr;  // the RIS driver (no cross-pixel reads: no halo, only the gather)
    rt_ctx ctx = nullptr;
    int W = 0, H = 0, iterations = 0, blur = 0;
    int rank = 0, world = 1;
    std::vector<int> bounds;
    Rccl::comm_t comm = nullptr;
    bool timing = false;
    // what the transfers of the plan being built read and write, for the library's log of stream operations (rt_debug_trace)
    struct RowsNote { bool send; rt_buffer buf; rt_texture tex; int r0, r1; };
    std::vector<RowsNote> notes;
    bool gatherPending = false;   // a frame has ended whose gather is not enqueued yet
    bool gatherStamped = false;   // MK_GATHER_DONE has been recorded
    double sums[RTH_MG_TIMES] = {};
    int timedFrames = 0;
    std::string err;
};

extern "C" {

// Row boundaries b[0..world] of the bands: multiples of 8 rows (the tile height) except the last, heights as equal as
// that allows.  Returns -1 if some band would be empty.
int rth_mg_band_bounds(int height, int world, int* out) {
    if (height < 1 || world < 1 || !out) return -1;
    for (int r = 0; r < world; r++) out[r] = (int)nearbyint((double)height * r / world / 8.0) * 8;  // round half to even
    out[world] = height;
    for (int r = 0; r < world; r++) if (out[r + 1] <= out[r]) return -1;
    return 0;
}
// The same with a cost per row: bands of roughly equal cost, none shorter than minRows (a band must be at least as tall
// as the halo its neighbours take from it).  The twin of multigpu.band_bounds(height, world, weights, min_rows).
int rth_mg_balanced_bounds(int height, int world, const double* rowCost, int minRows, int* out) {
    if (height < 1 || world < 1 || !out || !rowCost) return -1;
    minRows = (std::max(minRows, 8) + 7) / 8 * 8;
    std::vector<double> c((size_t)height + 1, 0.0);
    for (int i = 0; i < height; i++) c[(size_t)i + 1] = c[(size_t)i] + rowCost[i];
    out[0] = 0;
    for (int r = 1; r < world; r++) {
        const double want = c[(size_t)height] * r / world;
        const int row = (int)(std::lower_bound(c.begin(), c.end(), want) - c.begin());
        int cut = std::max(out[r - 1] + minRows, (int)nearbyint(row / 8.0) * 8);
        cut = std::min(cut, (height - (world - r) * minRows) / 8 * 8);  // leave room for the bands below
        out[r] = cut;
    }
    out[world] = height;
    for (int r = 0; r < world; r++) if (out[r + 1] <= out[r]) return -1;
    return 0;
}
// Per spatial pass i (MaxOffset = 3^i * blur, Main.cs:394): halo[i] = rows the pass reads beyond the rows it writes
// (MaxOffset down, MaxOffset up, one more up because a negative x offset wraps into the previous row), extend[i] =
// rows it must write beyond the band so that the later passes find their inputs.  Returns the rows exchanged once
// per frame (the sum of the halos).
int rth_mg_halo_plan(int iterations, int blur, int* halo, int* extend) {
    int total = 0;
    std::vector<int> h((size_t)(iterations > 0 ? iterations : 0));
    for (int i = 0; i < iterations; i++) { h[(size_t)i] = (int)(float)pow(3.0, (double)i) * blur + 1; total += h[(size_t)i]; }
    int later = 0;
    for (int i = iterations - 1; i >= 0; i--) {
        if (halo) halo[i] = h[(size_t)i];
        if (extend) extend[i] = later;
        later += h[(size_t)i];
    }
    return total;
}

int rth_mg_unique_id(unsigned char out[128]) {
    if (!out || !rccl().ok) return -1;
    Rccl::UniqueId id;
    if (rccl().GetUniqueId(&id) != 0) return -1;
    memcpy(out, id.internal, 128);
    return 0;
}

namespace {
bool mgBoundsOk(rth_mg g, const int* b, std::string& why) {
    bool ok = b[0] == 0 && b[g->world] == g->H;
    for (int r = 0; r < g->world; r++) ok = ok && b[r + 1] > b[r] && (b[r] % 8 == 0);
    if (!ok) { why = "band boundaries must start at 0, end at the image height, increase, and be multiples of 8"; return false; }
    const int total = rth_mg_halo_plan(g->iterations, g->blur, nullptr, nullptr);
    for (int r = 0; r < g->world; r++)
        if (g->world > 1 && b[r + 1] - b[r] < total) { why = "a band is shorter than the halo: use fewer GPUs or a larger image"; return false; }
    return true;
}
rth_mg mgCreate(rth_mg g, int rank, int world, const int* bounds, const unsigned char uniqueId[128], char* err, int errLen) {
    auto fail = [&](const std::string& msg) { if (err && errLen > 0) snprintf(err, (size_t)errLen, "%s", msg.c_str()); delete g; return (rth_mg) nullptr; };
    if (!g->ctx || world < 1 || rank < 0 || rank >= world) return fail("bad arguments");
    g->rank = rank; g->world = world;
    g->bounds.resize((size_t)world + 1);
    if (bounds) {  // the caller's cuts; every rank must pass the same ones
        for (int r = 0; r <= world; r++) g->bounds[(size_t)r] = bounds[r];
    } else if (rth_mg_band_bounds(g->H, world, g->bounds.data()) != 0) return fail("the image cannot be cut into that many bands of whole tiles");
    std::string why;
    if (!mgBoundsOk(g, g->bounds.data(), why)) return fail(why);
    // A communicator of one rank is legal (and lets a single GPU exercise the set-up).  Several ranks WITHOUT an id
    // get no communicator: such a rank can only be stepped with rth_mg_phase / rth_mg_plan, by a test that carries
    // the transfers out itself.
    if (uniqueId) {
        if (!rccl().ok) return fail(rccl().why);
        if (!rt_transfer_stream(g->ctx)) return fail(rt_last_error(g->ctx));
        Rccl::UniqueId id;
        memcpy(id.internal, uniqueId, 128);
        int rc = rccl().CommInitRank(&g->comm, world, id, rank);  // on the device current in this thread (rt_create made it so)
        if (rc != 0) return fail(std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc));
    }
    // this rank renders the same band every frame and never reads per-pixel data outside band and halo: its primary
    // rays may run beside the previous frame's kernels although they cover only part of the image
    rt_set_overlap(g->ctx, 2);
    return g;
}
}  // namespace

rth_mg rth_mg_create(rth_main m, int rank, int world, const int* bounds, const unsigned char uniqueId[128], char* err, int errLen) {
    rth_mg g = new rth_mg_s;
    if (m) {
        g->m = m; g->ctx = m->ctx;
        g->W = m->cfg.Resolution[0]; g->H = m->cfg.Resolution[1];
        g->iterations = m->cfg.SpatialReuseIterations; g->blur = m->cfg.SpatialReuseBlur;
    }
    return mgCreate(g, rank, world, bounds, uniqueId, err, errLen);
}
rth_mg rth_mg_create_ris(rth_ris_main m, int rank, int world, const int* bounds, const unsigned char uniqueId[128], char* err, int errLen) {
    rth_mg g = new rth_mg_s;
    int res[2] = {0, 0};
    if (m && rth_ris_main_resolution(m, res) == 0) {
        g->ris = m; g->ctx = rth_ris_main_ctx(m);
        g->W = res[0]; g->H = res[1];
    }
    return mgCreate(g, rank, world, bounds, uniqueId, err, errLen);
}
rth_hub rth_hub_create(int world) {
    if (world < 1 || !hipRt().ok) return nullptr;
    rth_hub_s* hub = new rth_hub_s;
    hub->world = world;
    hub->queue.resize((size_t)world * world);
    return hub;
}
void rth_hub_destroy(rth_hub hub) {
    if (!hub) return;
    for (void* ev : hub->spent) hipRt().EventDestroy(ev);   // (the ranks made with the hub are gone: their streams were waited for)
    delete hub;
}
int rth_mg_use_hub(rth_mg g, rth_hub hub) {
    if (!g || !hub) return -1;
    if (g->comm) { g->err = "this rank already has a communicator"; return -1; }
    if (hub->world != g->world) { g->err = "the hub was made for another number of ranks"; return -1; }
    if (!rt_transfer_stream(g->ctx)) { g->err = rt_last_error(g->ctx); return -1; }
    g->hub = hub;
    return 0;
}
void rth_mg_destroy(rth_mg g) {
    if (!g) return;
    if (g->comm) { rt_sync(g->ctx); rccl().CommDestroy(g->comm); }
    rt_sync(g->ctx);
    rt_set_row_band(g->ctx, 0, 0);
    rt_set_overlap(g->ctx, 1);
    delete g;
}
const char* rth_mg_last_error(rth_mg g) { return g ? g->err.c_str() : "null"; }
int rth_mg_bounds(rth_mg g, int* out) {
    if (!g || !out) return -1;
    for (size_t i = 0; i < g->bounds.size(); i++) out[i] = g->bounds[i];
    return 0;
}
int rth_mg_set_bounds(rth_mg g, const int* bounds) {
    if (!g || !bounds) return -1;
    std::string why;
    if (!mgBoundsOk(g, bounds, why)) { g->err = why; return -1; }
    for (int r = 0; r <= g->world; r++) g->bounds[(size_t)r] = bounds[r];
    return 0;
}

// Self-test of the transfer path on one rank: rows [src0, src1) of the Candidates buffer and of the depth texture are
// sent to, and received from, this same rank with the RCCL calls the frame loop uses (on the transfer stream, ordered
// after the kernels), landing at row dst0.
int rth_mg_loopback(rth_mg g, int src0, int src1, int dst0) {
    if (!g || !g->m || !g->comm || src1 <= src0) return -1;
    rth_main m = g->m;
    rt_ctx ctx = m->ctx;
    const int W = m->cfg.Resolution[0];
    m->CreateTextures();
    rt_segment a[5], b[5];
    int na = 0, nb = 0;
    if (rt_buffer_map_rows(m->CandidateBuffer.h, W, src0, src1, a, &na) < 0 || rt_texture_map_rows(m->DepthBufferTexture.h, src0, src1, &a[na]) < 0 ||
        rt_buffer_map_rows(m->CandidateBuffer.h, W, dst0, dst0 + (src1 - src0), b, &nb) < 0 ||
        rt_texture_map_rows(m->DepthBufferTexture.h, dst0, dst0 + (src1 - src0), &b[nb]) < 0 || na != nb) {
        g->err = rt_last_error(ctx);
        return -1;
    }
    void* stream = rt_transfer_stream(ctx);
    if (!stream || rt_stream_order(ctx, 0) < 0) { g->err = rt_last_error(ctx); return -1; }
    int rc = rccl().GroupStart();
    for (int k = 0; k <= na && rc == 0; k++) {
        rc = rccl().Send(a[k].ptr, a[k].bytes, NCCL_UINT8, g->rank, g->comm, stream);
        if (rc == 0) rc = rccl().Recv(b[k].ptr, b[k].bytes, NCCL_UINT8, g->rank, g->comm, stream);
    }
    int rc2 = rccl().GroupEnd();
    if (rc != 0 || rc2 != 0) { g->err = std::string("RCCL loopback: ") + rccl().GetErrorString(rc != 0 ? rc : rc2); return -1; }
    if (rt_stream_order(ctx, 1) < 0) { g->err = rt_last_error(ctx); return -1; }
    // (the frame loop never reads halo rows back; a caller of this self-test does)
    if (rt_buffer_rows_written(m->CandidateBuffer.h, W, dst0, dst0 + (src1 - src0)) < 0) { g->err = rt_last_error(ctx); return -1; }
    return 0;
}

}  // extern "C"

namespace {

void mgCheck(rth_mg g, int rc, const char* what) {
    if (rc < 0) throw DriverError{std::string(what) + ": " + rt_last_error(g->ctx)};
}
// Appends the transfers of rows [r0, r1) of a reservoir buffer and/or a texture to or from `peer`.
void mgRows(rth_mg g, std::vector<rth_xfer>& plan, bool send, rt_buffer buf, rt_texture tex, int r0, int r1, int peer) {
    if (r1 <= r0) return;
    rt_segment seg[5];
    int cnt = 0;
    g->notes.push_back({send, buf, tex, r0, r1});
    if (buf) mgCheck(g, rt_buffer_map_rows(buf, g->W, r0, r1, seg, &cnt), "rt_buffer_map_rows");
    if (tex) { mgCheck(g, rt_texture_map_rows(tex, r0, r1, &seg[cnt]), "rt_texture_map_rows"); cnt++; }
    for (int k = 0; k < cnt; k++) plan.push_back({send ? 1 : 0, peer, seg[k].ptr, seg[k].bytes});
}
// The one halo exchange of a frame: my border rows of the first spatial pass's source buffer and of the depth buffer go
// to both neighbours, theirs come here, `total` rows deep.
void mgPlanHalo(rth_mg g, std::vector<rth_xfer>& plan) {
    g->notes.clear();
    rth_main m = g->m;
    if (!m || g->iterations <= 0) return;
    const int rank = g->rank, world = g->world, H = g->H;
    const int b0 = g->bounds[(size_t)rank], b1 = g->bounds[(size_t)rank + 1];
    const int total = rth_mg_halo_plan(g->iterations, g->blur, nullptr, nullptr);
    rt_buffer cand = m->CandidateBuffer.h;
    rt_texture depth = m->DepthBufferTexture.h;
    if (rank > 0) {
        mgRows(g, plan, true, cand, depth, b0, b0 + total < b1 ? b0 + total : b1, rank - 1);
        mgRows(g, plan, false, cand, depth, b0 - total > 0 ? b0 - total : 0, b0, rank - 1);
    }
    if (rank < world - 1) {
        mgRows(g, plan, true, cand, depth, b1 - total > b0 ? b1 - total : b0, b1, rank + 1);
        mgRows(g, plan, false, cand, depth, b1, b1 + total < H ? b1 + total : H, rank + 1);
    }
}
// The displayed image is completed on rank 0.
void mgPlanGather(rth_mg g, std::vector<rth_xfer>& plan) {
    g->notes.clear();
    rt_texture result = g->m ? g->m->RTResultTexture.h : rth_ris_main_texture(g->ris, "Result");
    if (!result) throw DriverError{"the Result texture does not exist"};
    if (g->rank == 0) {
        for (int r = 1; r < g->world; r++) mgRows(g, plan, false, nullptr, result, g->bounds[(size_t)r], g->bounds[(size_t)r + 1], r);
    } else {
        mgRows(g, plan, true, nullptr, result, g->bounds[(size_t)g->rank], g->bounds[(size_t)g->rank + 1], 0);
    }
}
// The rows of the first spatial pass that read nothing a neighbour owns: the band shrunk by that pass's halo on every
// side that has a neighbour.
void mgInterior(rth_mg g, int* lo, int* hi) {
    const int b0 = g->bounds[(size_t)g->rank], b1 = g->bounds[(size_t)g->rank + 1];
    int halo0 = 0;
    if (g->iterations > 0) {
        std::vector<int> halo((size_t)g->iterations);
        rth_mg_halo_plan(g->iterations, g->blur, halo.data(), nullptr);
        halo0 = halo[0];
    }
    *lo = g->rank > 0 ? b0 + halo0 : b0;
    *hi = g->rank < g->world - 1 ? b1 - halo0 : b1;
    if (*hi < *lo) *hi = *lo;
}
// A frame in four steps.  0: everything before the halo exchange (InitialTrace, TemporalReuse on the band).
// 1: the first spatial pass on the interior rows (needs no halo).  2: the rest of the spatial reuse, which needs the
// halos: the first pass on the border rows, the later passes, TransferToOriginal.  3: TraceRays and AccumulateFrames
// (TraceRays writes this rank's band of Result, which the previous frame's gather may still be reading).
void mgPhase(rth_mg g, int phase) {
    rth_main m = g->m;
    rt_ctx ctx = g->ctx;
    const int rank = g->rank, world = g->world, H = g->H;
    const int b0 = g->bounds[(size_t)rank], b1 = g->bounds[(size_t)rank + 1];
    auto rows = [&](int r0, int r1) { mgCheck(g, rt_set_row_band(ctx, r0, r1), "rt_set_row_band"); };
    if (g->ris) {  // the RIS kernels read only their own pixel: no halo; TraceRays (with AccumulateFrames) is step 3
        auto ris = [&](int rc) { if (rc != 0) throw DriverError{rth_ris_main_last_error(g->ris)}; };
        rows(b0, b1);
        if (phase == 0) { ris(rth_ris_main_update_per_frame(g->ris)); ris(rth_ris_main_run_stage(g->ris, 0)); }
        if (phase == 3) { ris(rth_ris_main_run_stage(g->ris, 1)); ris(rth_ris_main_run_post_processing(g->ris)); }
        return;
    }
    m->CreateTextures();
    const int iterations = m->cfg.SpatialReuseIterations;
    int lo = b0, hi = b1;
    mgInterior(g, &lo, &hi);
    switch (phase) {
        case 0:
            rows(b0, b1);
            m->Update();
            m->RunStage(0);
            m->RunStage(1);
            return;
        case 1:
            if (iterations > 0 && hi > lo) { rows(lo, hi); m->SpatialReusePass(0); }
            return;
        case 2: {
            if (iterations <= 0) return;
            std::vector<int> halo((size_t)iterations), extend((size_t)iterations);
            rth_mg_halo_plan(iterations, m->cfg.SpatialReuseBlur, halo.data(), extend.data());
            for (int i = 0; i < iterations; i++) {
                // the earlier passes also write the rows the later passes will read from beyond the band
                int wlo = rank > 0 ? b0 - extend[(size_t)i] : b0, whi = rank < world - 1 ? b1 + extend[(size_t)i] : b1;
                wlo = wlo > 0 ? wlo : 0; whi = whi < H ? whi : H;
                if (i == 0) {  // the interior rows are done: the two border strips remain
                    if (lo > wlo) { rows(wlo, lo < whi ? lo : whi); m->SpatialReusePass(0); }
                    if (whi > hi) { rows(hi > wlo ? hi : wlo, whi); m->SpatialReusePass(0); }
                    if (hi <= lo && !(lo > wlo) && !(whi > hi)) { rows(wlo, whi); m->SpatialReusePass(0); }
                } else {
                    rows(wlo, whi);
                    m->SpatialReusePass(i);
                }
            }
            rows(b0, b1);
            m->TransferIfNeeded();
            return;
        }
        case 3:
            rows(b0, b1);
            m->RunStage(3);
            m->RunPostProcessingShader();
            return;
        default: throw DriverError{"unknown phase"};
    }
}
// Enqueues the transfers of `plan` as one RCCL group on the transfer stream.
void hubRun(rth_mg g, const std::vector<rth_xfer>& plan) {
    rth_hub_s* hub = g->hub;
    HipRt& hip = hipRt();
    void* stream = rt_transfer_stream(g->ctx);
    if (!stream) throw DriverError{rt_last_error(g->ctx)};
    auto check = [&](int rc, const char* what) { if (rc != 0) { { std::lock_guard<std::mutex> l(hub->mu); hub->broken = true; } hub->cv.notify_all(); throw DriverError{std::string(what) + " failed in the rehearsal transport"}; } };
    const int me = g->rank, world = hub->world;
    std::vector<std::shared_ptr<rth_hub_s::SendRec>> mine;
    // the sends first: what precedes this group on the transfer stream is what the receiver has to wait for
    for (const auto& x : plan) {
        if (!x.send) continue;
        auto rec = std::make_shared<rth_hub_s::SendRec>();
        rec->ptr = x.ptr; rec->bytes = x.bytes;
        check(hip.EventCreateWithFlags(&rec->ready, 2 /* hipEventDisableTiming */), "hipEventCreate");
        check(hip.EventRecord(rec->ready, stream), "hipEventRecord");
        { std::lock_guard<std::mutex> l(hub->mu); hub->queue[(size_t)me * world + x.peer].push_back(rec); }
        mine.push_back(rec);
    }
    hub->cv.notify_all();
    for (const auto& x : plan) {
        if (x.send) continue;
        std::shared_ptr<rth_hub_s::SendRec> rec;
        {
            std::unique_lock<std::mutex> l(hub->mu);
            auto& q = hub->queue[(size_t)x.peer * world + me];
            hub->cv.wait(l, [&] { return hub->broken || !q.empty(); });
            if (hub->broken) throw DriverError{"another rank of the rehearsal gave up"};
            rec = q.front();
            q.pop_front();
        }
        if (rec->bytes != x.bytes) check(1, "matching a send of another size");
        check(hip.StreamWaitEvent(stream, rec->ready, 0), "hipStreamWaitEvent");
        check(hip.MemcpyAsync(x.ptr, rec->ptr, x.bytes, 3 /* hipMemcpyDeviceToDevice */, stream), "hipMemcpyAsync");
        void* copied = nullptr;
        check(hip.EventCreateWithFlags(&copied, 2), "hipEventCreate");
        check(hip.EventRecord(copied, stream), "hipEventRecord");
        { std::lock_guard<std::mutex> l(hub->mu); rec->copied = copied; rec->taken = true; }
        hub->cv.notify_all();
    }
    for (auto& rec : mine) {   // the sender's stream goes on when its rows have been copied out
        {
            std::unique_lock<std::mutex> l(hub->mu);
            hub->cv.wait(l, [&] { return hub->broken || rec->taken; });
            if (hub->broken) throw DriverError{"another rank of the rehearsal gave up"};
        }
        check(hip.StreamWaitEvent(stream, rec->copied, 0), "hipStreamWaitEvent");
        std::lock_guard<std::mutex> l(hub->mu);
        hub->spent.push_back(rec->ready);
        hub->spent.push_back(rec->copied);
    }
}
void mgRunTransfers(rth_mg g, const std::vector<rth_xfer>& plan);
// The transfers of the plan just built (mgPlanHalo / mgPlanGather), enqueued on the transfer stream.
void mgRun(rth_mg g, const std::vector<rth_xfer>& plan) {
    if (plan.empty()) return;
    mgRunTransfers(g, plan);
    for (const auto& n : g->notes) mgCheck(g, rt_debug_note_rows(g->ctx, 1, n.buf, n.tex, n.r0, n.r1, n.send ? 0 : 1), "rt_debug_note_rows");
}
void mgRunTransfers(rth_mg g, const std::vector<rth_xfer>& plan) {
    if (g->hub) return hubRun(g, plan);
    auto nccl = [&](int rc, const char* what) { if (rc != 0) throw DriverError{std::string(what) + ": " + rccl().GetErrorString(rc)}; };
    void* stream = rt_transfer_stream(g->ctx);
    if (!stream) throw DriverError{rt_last_error(g->ctx)};
    nccl(rccl().GroupStart(), "ncclGroupStart");
    for (const auto& x : plan) {
        if (x.send) nccl(rccl().Send(x.ptr, x.bytes, NCCL_UINT8, x.peer, g->comm, stream), "ncclSend");
        else nccl(rccl().Recv(x.ptr, x.bytes, NCCL_UINT8, x.peer, g->comm, stream), "ncclRecv");
    }
    nccl(rccl().GroupEnd(), "ncclGroupEnd");
}
void mgMark(rth_mg g, int stream, int slot) {
    if (g->timing) mgCheck(g, rt_mark(g->ctx, stream, slot), "rt_mark");
}
void mgOrder(rth_mg g, int first) { mgCheck(g, rt_stream_order(g->ctx, first), "rt_stream_order"); }
double mgElapsed(rth_mg g, int a, int b) {
    double ms = 0;
    mgCheck(g, rt_mark_elapsed(g->ctx, a, b, &ms), "rt_mark_elapsed");
    return ms;
}

}  // namespace

extern "C" {

// n banded frames, enqueued (asynchronous; rt_sync waits for them, all streams).
// The library runs consecutive frames on alternating dispatch streams (two frames in flight), so the order of the
// transfers decides how much of frame f+1 can run beside frame f: the gather of frame f is enqueued BEHIND the halo
// exchange of frame f+1 (every rank enqueues halo 0, halo 1, gather 0, halo 2, gather 1, ... and the last gather at the
// end of the call), and each wait is for a stamp, not for the present end of a stream:
//   halo f      waits for InitialTrace + TemporalReuse of frame f (rt_stream_order)
//   gather f-1  waits for the stamp at the end of frame f-1
//   frame f     waits for halo f before the border rows of its first pass, and for gather f-1 before its TraceRays
//               (which overwrites the band of Result the gather reads).
// With the per-step timing on, a frame is waited for before the next starts, and the gather follows its own frame.
int rth_mg_render_frames(rth_mg g, int n) {
    if (!g) return -1;
    if (g->world > 1 && !g->comm && !g->hub) { g->err = "this rank was created without a communicator"; return -1; }
    try {
        std::vector<rth_xfer> plan;
        // (a communicator of one rank has nothing to transfer, but takes the same path through the loop)
        const bool multi = g->world > 1 || g->comm || g->hub;
        // (RTH_MG_PIPELINE=0 in the environment - of every rank - keeps the gather of a frame ahead of the next frame's halo
        // exchange, as with the per-step timing: the order for narrowing down a problem on new hardware)
        static const bool pipelineAllowed = [] { const char* e = getenv("RTH_MG_PIPELINE"); return e ? atoi(e) != 0 : true; }();
        const bool pipelined = multi && !g->timing && pipelineAllowed;
        auto gather = [&] {
            plan.clear();
            mgPlanGather(g, plan);
            mgRun(g, plan);
        };
        for (int f = 0; f < n; f++) {
            mgMark(g, 0, MK_FRAME);
            mgPhase(g, 0);
            mgMark(g, 0, MK_PRE);
            if (multi) {  // the halos leave once the band's reservoirs exist, and travel beside the interior rows
                plan.clear();
                mgPlanHalo(g, plan);
                mgOrder(g, 0);
                mgMark(g, 1, MK_HALO0);
                mgRun(g, plan);
                mgMark(g, 1, MK_HALO1);
                if (pipelined) {
                    mgCheck(g, rt_mark(g->ctx, 1, MK_HALO_DONE), "rt_mark");
                    if (g->gatherPending) {   // the frame before: its gather travels beside this frame's reuse passes
                        mgCheck(g, rt_wait_mark(g->ctx, 1, MK_FRAME_END), "rt_wait_mark");
                        gather();
                        mgCheck(g, rt_mark(g->ctx, 1, MK_GATHER_DONE), "rt_mark");
                        g->gatherPending = false;
                        g->gatherStamped = true;
                    }
                }
            }
            mgPhase(g, 1);
            mgMark(g, 0, MK_INTERIOR);
            if (pipelined) mgCheck(g, rt_wait_mark(g->ctx, 0, MK_HALO_DONE), "rt_wait_mark");
            else if (multi) mgOrder(g, 1);
            mgPhase(g, 2);
            mgMark(g, 0, MK_POST0);
            // TraceRays overwrites this rank's band of Result: not before the last gather has read it (without the
            // pipelining, mgOrder above already made the dispatch stream wait for everything on the transfer stream)
            if (pipelined && g->gatherStamped) mgCheck(g, rt_wait_mark(g->ctx, 0, MK_GATHER_DONE), "rt_wait_mark");
            mgPhase(g, 3);
            mgMark(g, 0, MK_END);
            if (pipelined) {
                mgCheck(g, rt_mark(g->ctx, 0, MK_FRAME_END), "rt_mark");
                g->gatherPending = true;
            } else if (multi) {
                mgOrder(g, 0);
                mgMark(g, 1, MK_GATHER0);
                gather();
                mgMark(g, 1, MK_GATHER1);
            }
            if (g->timing) {  // reading the stamps waits for the frame: timing mode gives up the overlap between frames
                g->sums[RTH_MG_T_FRAME] += mgElapsed(g, MK_FRAME, MK_END);
                g->sums[RTH_MG_T_BEFORE_HALO] += mgElapsed(g, MK_FRAME, MK_PRE);
                g->sums[RTH_MG_T_INTERIOR] += mgElapsed(g, MK_PRE, MK_INTERIOR);
                g->sums[RTH_MG_T_AFTER_HALO] += mgElapsed(g, MK_INTERIOR, MK_END);
                if (multi) {
                    g->sums[RTH_MG_T_HALO] += mgElapsed(g, MK_HALO0, MK_HALO1);
                    g->sums[RTH_MG_T_GATHER] += mgElapsed(g, MK_GATHER0, MK_GATHER1);
                    // how long the dispatch stream stood still for the halos: from the end of the interior rows to
                    // the arrival of the last halo row, if that came later
                    const double wait = mgElapsed(g, MK_INTERIOR, MK_HALO1);
                    g->sums[RTH_MG_T_HALO_EXPOSED] += wait > 0 ? wait : 0;
                }
                g->timedFrames++;
            }
        }
        if (g->gatherPending) {   // the last frame of this call: every frame asked for is complete on rank 0 when the call's work is
            mgCheck(g, rt_wait_mark(g->ctx, 1, MK_FRAME_END), "rt_wait_mark");
            gather();
            mgCheck(g, rt_mark(g->ctx, 1, MK_GATHER_DONE), "rt_mark");
            g->gatherPending = false;
            g->gatherStamped = true;
        }
    } catch (const DriverError& e) {
        g->err = e.msg;
        return -1;
    }
    return 0;
}
int rth_mg_enable_timing(rth_mg g, int on) {
    if (!g) return -1;
    g->timing = on != 0;
    for (double& s : g->sums) s = 0;
    g->timedFrames = 0;
    return 0;
}
int rth_mg_times(rth_mg g, double out[RTH_MG_TIMES]) {
    if (!g || !out) return -1;
    for (int i = 0; i < RTH_MG_TIMES; i++) out[i] = g->timedFrames ? g->sums[i] / g->timedFrames : 0.0;
    return g->timedFrames;
}
// The steps of a frame and the transfer lists between them, one at a time: lets a test that holds several ranks in one
// process (all on one GPU, no communicator) carry the transfers out itself and so check the schedule.
int rth_mg_phase(rth_mg g, int phase) {
    if (!g || phase < 0 || phase > 3) return -1;
    try { mgPhase(g, phase); } catch (const DriverError& e) { g->err = e.msg; return -1; }
    return 0;
}
int rth_mg_plan(rth_mg g, int which, rth_xfer* out, int cap) {
    if (!g || which < 0 || which > 1) return -1;
    std::vector<rth_xfer> plan;
    try {
        if (g->m) g->m->CreateTextures();
        if (which == 0) mgPlanHalo(g, plan); else mgPlanGather(g, plan);
    } catch (const DriverError& e) { g->err = e.msg; return -1; }
    if ((int)plan.size() > cap || (!out && !plan.empty())) return -1;
    for (size_t i = 0; i < plan.size(); i++) out[i] = plan[i];
    return (int)plan.size();
}

// Cost-balanced bands without a frame rendered in full: this rank renders the primary rays of frame 0 on its present
// band (what the real frame 0 will render again, so nothing is disturbed), reads the band's depth back and writes
// 1 + 24 per hit pixel for each of its rows into rowCost[H] (0 for the rows of other ranks).  The frame counters are
// put back, so the first rth_mg_render_frames still renders frame 0.
int rth_mg_row_costs(rth_mg g, double* rowCost) {
    if (!g || !rowCost || !g->m) return -1;
    rth_main m = g->m;
    try {
        const int b0 = g->bounds[(size_t)g->rank], b1 = g->bounds[(size_t)g->rank + 1];
        m->CreateTextures();
        mgCheck(g, rt_set_row_band(g->ctx, b0, b1), "rt_set_row_band");
        const int fc = m->FrameCount, fs = m->framesStarted;
        m->UpdatePerFrame();
        m->RunStage(0);
        m->FrameCount = fc; m->framesStarted = fs;
        std::vector<float> depth((size_t)(b1 - b0) * g->W);
        mgCheck(g, rt_texture_get_rows(m->DepthBufferTexture.h, depth.data(), b0, b1), "rt_texture_get_rows");
        for (int y = 0; y < g->H; y++) rowCost[y] = 0.0;
        for (int y = b0; y < b1; y++) {
            int hits = 0;
            for (int x = 0; x < g->W; x++) hits += std::isfinite(depth[(size_t)(y - b0) * g->W + x]) ? 1 : 0;
            rowCost[y] = (double)g->W + 24.0 * hits;   // multigpu.row_cost
        }
    } catch (const DriverError& e) { g->err = e.msg; return -1; }
    return 0;
}
// Collective: every rank measures its rows (rth_mg_row_costs), the vectors are summed over the ranks with one
// ncclAllReduce of H doubles, and every rank derives the same balanced cuts from the sum.
int rth_mg_balance(rth_mg g) {
    if (!g || !g->m) return -1;
    if (g->world == 1) return 0;
    if (!g->comm) { g->err = "this rank was created without a communicator"; return -1; }
    // A collective: a rank that fails on its own must still join the all-reduce, or the others wait for it for ever.
    // So the buffer the all-reduce works on is made first (a failure there is the one case this rank cannot report
    // through the collective), and the measurement's outcome travels in one extra element: every rank sees the number
    // of ranks that failed and all of them give up together.
    const size_t n = (size_t)g->H + 1;
    std::vector<double> cost(n, 0.0);
    rt_buffer buf = nullptr;
    rt_segment seg[4];
    int cnt = 0;
    void* stream = rt_transfer_stream(g->ctx);
    if (!stream || rt_buffer_create(g->ctx, (int64_t)n, 8, &buf) < 0 || rt_buffer_map_rows(buf, 1, 0, (int)n, seg, &cnt) < 0) {
        g->err = std::string("rth_mg_balance could not set up its exchange buffer: ") + rt_last_error(g->ctx);
        rt_buffer_release(buf);
        return -1;
    }
    std::string localFailure;
    if (rth_mg_row_costs(g, cost.data()) != 0) { localFailure = g->err; cost.assign(n, 0.0); cost[n - 1] = 1.0; }
    int rc = -1;
    try {
        if (rt_buffer_set_data(buf, cost.data(), n * 8) < 0) {   // (the upload failed: join with whatever the buffer holds, flagged below)
            if (localFailure.empty()) localFailure = std::string("rt_buffer_set_data: ") + rt_last_error(g->ctx);
        }
        mgOrder(g, 0);
        int nrc = rccl().AllReduce(seg[0].ptr, seg[0].ptr, n, NCCL_DOUBLE, NCCL_SUM, g->comm, stream);
        if (nrc != 0) throw DriverError{std::string("ncclAllReduce: ") + rccl().GetErrorString(nrc)};
        mgOrder(g, 1);
        mgCheck(g, rt_buffer_get_data(buf, cost.data(), n * 8), "rt_buffer_get_data");
        if (!localFailure.empty()) throw DriverError{localFailure};
        if (cost[n - 1] != 0.0) throw DriverError{"another rank could not measure its rows: no rank changes its cuts"};
        std::vector<int> b((size_t)g->world + 1);
        const int total = rth_mg_halo_plan(g->iterations, g->blur, nullptr, nullptr);
        if (rth_mg_balanced_bounds(g->H, g->world, cost.data(), total, b.data()) != 0) throw DriverError{"no balanced cuts exist for this image"};
        rc = rth_mg_set_bounds(g, b.data());
    } catch (const DriverError& e) { g->err = e.msg; rc = -1; }
    rt_buffer_release(buf);
    return rc;
}

}  // extern "C"
