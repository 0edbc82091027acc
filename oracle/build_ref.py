"""Builds oracle/_ref/libref_rasterizer.so: the REFERENCE's own rasterizer sources
(/root/reference/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu) and simple-knn
(third_party/simple-knn/simple_knn.cu) and the __global__ kernels of src/operate_points.cu / src/stereo_vision.cu,
compiled for the host with g++ against the
CUDA / cooperative-groups / CUB / glm shims in oracle/ref_shim/ (our code: a fiber model of thread blocks and the
published semantics of the two CUB calls and of the glm operators the sources use).

TEST INFRASTRUCTURE ONLY.  It exists to PIN the CPU oracle (oracle/gsr_oracle.c) and, through committed fixtures
(tests/golden/reference_small.npz, made by tests/golden/make_reference_golden.py), the HIP kernels against the
reference's own code.  Nothing from /root/reference is copied into the repository: the sources are read where they lie;
the only rewrite is the kernel-launch syntax `k<<<g, b>>>(args)` -> `CUDAEMU_LAUNCH((k), g, b, args)`, done on the fly
into oracle/_ref/gen/ (git-ignored).  The reference tree is absent on the GPU boxes: there only the built .so or the
fixtures are used.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GSR_REFERENCE_ROOT", "/root/reference")
SRC_DIR = os.path.join(REF, "cuda_rasterizer")
OUT_DIR = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT_DIR, "gen")
OUT = os.path.join(OUT_DIR, "libref_rasterizer.so")
SOURCES = ["forward.cu", "backward.cu", "rasterizer_impl.cu"]
KNN_DIR = os.path.join(REF, "third_party", "simple-knn")   # simple_knn.cu: the distance initialisation (distCUDA2)
LAUNCH = re.compile(r"(\b\w+(?:<[^<>;(){}]*>)?)\s*<<\s*<\s*(.+?)\s*>>\s*>\s*\(")


POINT_SRC = [os.path.join(REF, "src", "operate_points.cu"), os.path.join(REF, "src", "stereo_vision.cu")]


def _global_functions(text):
    """The `__global__` kernel definitions of a .cu file, verbatim (the files also hold LibTorch wrappers, which need torch)."""
    out, i = [], 0
    while True:
        i = text.find("__global__", i)
        if i < 0:
            return out
        j = text.index("{", i)
        depth, k = 1, j + 1
        while depth:
            depth += {"{": 1, "}": -1}.get(text[k], 0)
            k += 1
        out.append(text[i:k])
        i = k


def available():
    return all(os.path.exists(os.path.join(SRC_DIR, s)) for s in SOURCES) and os.path.exists(os.path.join(KNN_DIR, "simple_knn.cu")) \
        and all(os.path.exists(f) for f in POINT_SRC)


def build(force=False):
    if not available():
        return OUT if os.path.exists(OUT) else None
    deps = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)] + POINT_SRC + [os.path.join(KNN_DIR, "simple_knn.cu"),
                                                                     os.path.join(HERE, "ref_api.cpp"), __file__]
    for root, _, files in os.walk(os.path.join(HERE, "ref_shim")):
        deps += [os.path.join(root, f) for f in files]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(GEN, exist_ok=True)
    gen = []
    for s in SOURCES + ["simple_knn.cu"]:
        src_dir = KNN_DIR if s == "simple_knn.cu" else SRC_DIR
        text = open(os.path.join(src_dir, s)).read()
        text, n = LAUNCH.subn(lambda m: f"CUDAEMU_LAUNCH(({m.group(1)}), {m.group(2)}, ", text)
        dst = os.path.join(GEN, s.replace(".cu", ".cpp"))
        with open(dst, "w") as f:
            f.write(f'#line 1 "{os.path.join(src_dir, s)}"\n' + text)
        gen.append(dst)
    # Photo-SLAM's point kernels: the __global__ functions of src/operate_points.cu and src/stereo_vision.cu (their device
    # helpers live in cuda_rasterizer/operate_points.h and stereo_vision.h, included as they are)
    kernels = []
    for f in POINT_SRC:
        kernels += _global_functions(open(f).read())
    dst = os.path.join(GEN, "point_kernels.cpp")
    with open(dst, "w") as f:
        f.write('#include <cuda_runtime.h>\n#include <cooperative_groups.h>\nnamespace cg = cooperative_groups;\n'
                '#include "operate_points.h"\n#include "stereo_vision.h"\n\n' + "\n\n".join(kernels) + "\n")
    gen.append(dst)
    shim = os.path.join(HERE, "ref_shim")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-w",
           "-I", os.path.join(shim, "include"), "-I", SRC_DIR, "-I", KNN_DIR, "-I", shim,
           "-o", OUT] + gen + [os.path.join(shim, "cudaemu.cpp"), os.path.join(HERE, "ref_api.cpp")]
    subprocess.check_call(cmd)
    import shutil
    shutil.rmtree(GEN, ignore_errors=True)   # the rewritten copies of the reference sources do not outlive the compile
    return OUT


LOSS_OUT = os.path.join(OUT_DIR, "libref_loss.so")
LOSS_HEADER = os.path.join(REF, "include", "loss_utils.h")


def build_loss(force=False):
    """oracle/_ref/libref_loss.so: torch ops (torch.ops.photoslam_reference.*) around the reference's include/loss_utils.h."""
    if not os.path.exists(LOSS_HEADER):
        return LOSS_OUT if os.path.exists(LOSS_OUT) else None
    src = os.path.join(HERE, "ref_loss.cpp")
    if not force and os.path.exists(LOSS_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(LOSS_OUT) for d in (src, LOSS_HEADER, __file__)):
        return LOSS_OUT
    import torch
    base = os.path.dirname(torch.__file__)
    inc = [os.path.join(base, "include"), os.path.join(base, "include", "torch", "csrc", "api", "include")]
    libdir = os.path.join(base, "lib")
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                           "-w", "-I" + os.path.join(REF, "include")] + ["-I" + i for i in inc] +
                          [src, "-o", LOSS_OUT, "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,-rpath," + libdir])
    return LOSS_OUT


MODEL_SRC = os.path.join(REF, "src", "gaussian_model.cpp")
MODEL_OUT = {"cpu": os.path.join(OUT_DIR, "libref_densify.so"), "cuda": os.path.join(OUT_DIR, "libref_densify_cuda.so")}
# the member functions of GaussianModel that oracle/ref_densify.cpp compiles, extracted verbatim by name
MODEL_FUNCTIONS = ["getScalingActivation", "getXYZ", "getOpacityActivation", "trainingSetup", "resetOpacity",
                   "replaceTensorToOptimizer", "prunePoints", "densificationPostfix", "densifyAndSplit", "densifyAndClone",
                   "densifyAndPrune", "addDensificationStats", "percentDense", "setPercentDense", "loadPly", "savePly", "increasePcd"]
ADAM_KEY = "c10::guts::to_string(param.unsafeGetTensorImpl())"   # LibTorch <= 2.1 state key (src/gaussian_model.cpp:571,598,670)


def _member_function(text, name):
    """`<return type> GaussianModel::name(...) {...}` of a .cpp file, verbatim -- every overload of the name."""
    out = []
    for m in re.finditer(r"^[\w:<>&\* ]+\bGaussianModel::" + name + r"\s*\(", text, re.M):
        j = text.index("{", m.end())
        depth, k = 1, j + 1
        while depth:
            depth += {"{": 1, "}": -1}.get(text[k], 0)
            k += 1
        out.append(text[m.start():k])
    if not out:
        raise RuntimeError(f"GaussianModel::{name} not found in {MODEL_SRC}")
    return "\n\n".join(out)


def build_densify(force=False):
    """oracle/_ref/libref_densify.so (host) and libref_densify_cuda.so (GPU boxes): torch ops around the reference's own
    densification / Adam-state code, see oracle/ref_densify.cpp.  Returns {"cpu": path, "cuda": path} (None where absent)."""
    src = os.path.join(HERE, "ref_densify.cpp")
    have = {k: (v if os.path.exists(v) else None) for k, v in MODEL_OUT.items()}
    if not os.path.exists(MODEL_SRC):
        return have
    deps = (src, MODEL_SRC, os.path.join(REF, "third_party", "tinyply", "tinyply.h"),
            os.path.join(REF, "include", "general_utils.h"), os.path.join(REF, "include", "gaussian_parameters.h"),
            os.path.join(REF, "src", "gaussian_parameters.cpp"), __file__)
    if not force and all(have.values()) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps for o in MODEL_OUT.values()):
        return have
    import shutil
    import torch
    text = open(MODEL_SRC).read()
    body = "\n\n".join(_member_function(text, n) for n in MODEL_FUNCTIONS)
    assert body.count(ADAM_KEY) == 6, "the Adam state key idiom changed in the reference"
    assert body.count("GaussianModel::increasePcd(") == 2, "two overloads of increasePcd expected"
    body = body.replace(ADAM_KEY, "param.unsafeGetTensorImpl()")
    os.makedirs(GEN, exist_ok=True)
    with open(os.path.join(GEN, "ref_gaussian_model_functions.inc"), "w") as f:
        f.write(f'#line 1 "{MODEL_SRC} (extract)"\n' + body + "\n")
    base = os.path.dirname(torch.__file__)
    inc = [os.path.join(base, "include"), os.path.join(base, "include", "torch", "csrc", "api", "include")]
    libdir = os.path.join(base, "lib")
    try:
        for kind, out in MODEL_OUT.items():
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                                   "-w", "-I" + GEN, "-I" + os.path.join(REF, "include"), "-I" + REF] + ["-I" + i for i in inc] +
                                  (["-DREF_DENSIFY_DEVICE_CPU"] if kind == "cpu" else []) +
                                  [src, os.path.join(REF, "src", "gaussian_parameters.cpp"), "-o", out, "-L" + libdir, "-ltorch",
                                   "-ltorch_cpu", "-lc10", "-ldl", "-Wl,-rpath," + libdir])
    finally:
        shutil.rmtree(GEN, ignore_errors=True)   # the extracted reference text does not outlive the compile
    return dict(MODEL_OUT)


LINK_OUT = {"emu": os.path.join(OUT_DIR, "ref_link_consumer_emu"), "hip": os.path.join(OUT_DIR, "ref_link_consumer_hip")}


def build_link_consumer(force=False):
    """oracle/_ref/ref_link_consumer_{emu,hip}: tests/ref_link/consumer.cpp compiled against the REFERENCE's declarations
    (include/rasterize_points.h as it is; include/gaussian_rasterizer.h without its `#include "gaussian_model.h"` line,
    written to oracle/_ref/gen/ and deleted after the compile) and linked against this repository's host library -- the
    emulator build for this container, the HIP build for the GPU boxes (where the reference tree does not exist: the
    prebuilt binary travels).  Returns {"emu": path, "hip": path} (None where absent)."""
    have = {k: (v if os.path.exists(v) else None) for k, v in LINK_OUT.items()}
    hdr = os.path.join(REF, "include", "gaussian_rasterizer.h")
    if not os.path.exists(hdr):
        return have
    import shutil
    import torch
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "photo-slam_amd", "host"))
    import build_host
    src = os.path.join(root, "tests", "ref_link", "consumer.cpp")
    build_host.build("emu")
    build_host.build("hip")
    # the boundary libraries (the reference's `cuda_rasterizer` / `simple_knn` by name) + this repository's GaussianRasterizer
    libs = {k: list(build_host.outputs(k).values()) for k in ("emu", "hip")}
    deps = [src, hdr, os.path.join(REF, "include", "rasterize_points.h"), __file__] + libs["emu"] + libs["hip"]
    if not force and all(have.values()) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps for o in LINK_OUT.values()):
        return have
    text = open(hdr).read()
    assert text.count('#include "gaussian_model.h"') == 1
    os.makedirs(GEN, exist_ok=True)
    with open(os.path.join(GEN, "gaussian_rasterizer.h"), "w") as f:
        f.write(f'#line 1 "{hdr}"\n' + text.replace('#include "gaussian_model.h"', "/* gaussian_model.h: not needed by these declarations */"))
    base = os.path.dirname(torch.__file__)
    inc = [os.path.join(base, "include"), os.path.join(base, "include", "torch", "csrc", "api", "include")]
    libdir = os.path.join(base, "lib")
    try:
        for kind, out in LINK_OUT.items():
            extra = ["-ltorch_hip", "-lc10_hip"] if kind == "hip" else []
            dirs = sorted({os.path.dirname(l) for l in libs[kind]})
            # GEN first: "gaussian_rasterizer.h" resolves to the generated copy, "rasterize_points.h" to the reference's own file
            subprocess.check_call(["g++", "-std=c++17", "-O1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
                                   "-I" + GEN, "-I" + os.path.join(REF, "include")] + ["-I" + i for i in inc] +
                                  # LibTorch FIRST: a pip wheel's bundled ROCm runtime (torch/lib/libamdhip64.so, SONAME
                                  # libamdhip64.so.7) then satisfies libgsr_hip.so's request for libamdhip64.so.7 -- one runtime
                                  ["-Wl,--no-as-needed", src, "-o", out, "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10"] + extra +
                                  ["-L" + d for d in dirs] + ["-l" + os.path.basename(l)[3:-3] for l in libs[kind]] +
                                  ["-Wl,-rpath," + libdir] + ["-Wl,-rpath," + d for d in dirs])
    finally:
        shutil.rmtree(GEN, ignore_errors=True)
    return dict(LINK_OUT)


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's HOST code, unchanged, on this repository's kernels.

HOST_OUT = {"emu": os.path.join(OUT_DIR, "libref_host_emu.so"), "hip": os.path.join(OUT_DIR, "libref_host_hip.so"),
            # the same sources with ONE header swapped: include/loss_utils.h -> this repository's host/include/loss_utils.h (same
            # names and signatures, l1_loss / ssim on the fused HIP kernels) -- what a maintainer gets who also swaps that header
            "emu_fused_loss": os.path.join(OUT_DIR, "libref_host_emu_fused_loss.so"),
            "hip_fused_loss": os.path.join(OUT_DIR, "libref_host_hip_fused_loss.so")}
HOST_OPS = {"emu": "photoslam_reference_host_emu", "hip": "photoslam_reference_host",
            "emu_fused_loss": "photoslam_reference_host_emu_fl", "hip_fused_loss": "photoslam_reference_host_fl"}
# compiled VERBATIM (the files themselves, through a tree of symbolic links -- no copy, no rewrite)
HOST_SOURCES = ["gaussian_rasterizer.cpp", "gaussian_renderer.cpp", "gaussian_trainer.cpp", "gaussian_parameters.cpp"]
HOST_HEADERS_REF = ["gaussian_renderer.h", "gaussian_rasterizer.h", "rasterize_points.h", "operate_points.h", "loss_utils.h",
                    "sh_utils.h", "general_utils.h", "gaussian_parameters.h", "types.h"]
# OUR stand-ins for the four headers that need Sophus / Eigen / OpenCV / ORB-SLAM3 (oracle/ref_host/)
HOST_HEADERS_STANDIN = ["gaussian_model.h", "gaussian_keyframe.h", "gaussian_scene.h", "gaussian_trainer.h", "sophus_standin.h"]
# every member function oracle/ref_host/gaussian_model.h declares, extracted verbatim by name from src/gaussian_model.cpp
HOST_MODEL_FUNCTIONS = ["getScalingActivation", "getRotationActivation", "getXYZ", "getFeatures", "getOpacityActivation",
                        "getCovarianceActivation", "oneUpShDegree", "setShDegree", "increasePcd", "applyScaledTransformation",
                        "scaledTransformationPostfix",
                        "scaledTransformVisiblePointsOfKeyframe", "trainingSetup", "updateLearningRate", "setPositionLearningRate",
                        "setFeatureLearningRate", "setOpacityLearningRate", "setScalingLearningRate", "setRotationLearningRate",
                        "resetOpacity", "replaceTensorToOptimizer", "prunePoints", "densificationPostfix", "densifyAndSplit",
                        "densifyAndClone", "densifyAndPrune", "addDensificationStats", "loadPly", "savePly", "percentDense",
                        "setPercentDense", "exponLrFunc"]


def build_host_tree(force=False):
    """oracle/_ref/libref_host_{emu,hip}.so: the reference's src/gaussian_rasterizer.cpp, src/gaussian_renderer.cpp,
    src/gaussian_trainer.cpp and src/gaussian_parameters.cpp compiled VERBATIM -- `g++ -c` on the files themselves, reached
    through a tree of symbolic links in which only gaussian_model.h / gaussian_keyframe.h / gaussian_scene.h /
    gaussian_trainer.h resolve to the stand-ins of oracle/ref_host/ -- together with the member functions of GaussianModel
    extracted by name from src/gaussian_model.cpp (no rewrite: host/include/compat/ supplies the two LibTorch drifts) and
    the glue of oracle/ref_host.cpp.  `hip`: linked against photo-slam_amd/lib/libcuda_rasterizer.so + libsimple_knn.so (the
    CMake targets named like the reference's) and nothing else of this repository; `emu`: against
    tests/emu/libcuda_rasterizer_emu.so with oracle/ref_host/emu_device.h force-included.
    `*_fused_loss`: the same with include/loss_utils.h resolving to this repository's host/include/loss_utils.h.
    Returns {flavour: path} (None where absent); torch op namespaces: HOST_OPS."""
    have = {k: (v if os.path.exists(v) else None) for k, v in HOST_OUT.items()}
    if not os.path.exists(MODEL_SRC):
        return have
    import shutil
    import torch
    root = os.path.dirname(HERE)
    host_dir = os.path.join(root, "photo-slam_amd", "host")
    sys.path.insert(0, host_dir)
    import build_host
    build_host.build("emu")
    build_host.build("hip")
    libs = {"emu": [build_host.outputs("emu")["cuda_rasterizer"]],
            "hip": [build_host.outputs("hip")["cuda_rasterizer"], build_host.outputs("hip")["simple_knn"]]}
    glue = os.path.join(HERE, "ref_host.cpp")
    standin = os.path.join(HERE, "ref_host")
    compat = os.path.join(host_dir, "include", "compat")
    deps = [glue, MODEL_SRC, __file__] + [os.path.join(standin, f) for f in os.listdir(standin)] + \
        [os.path.join(REF, "src", f) for f in HOST_SOURCES] + [os.path.join(REF, "include", f) for f in HOST_HEADERS_REF] + \
        [os.path.join(r, f) for r, _, fs in os.walk(compat) for f in fs] + libs["emu"] + libs["hip"] + [os.path.join(host_dir, "include", "loss_utils.h")]
    if not force and all(have.values()) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps for o in HOST_OUT.values()):
        return have
    our_loss = os.path.join(host_dir, "include", "loss_utils.h")
    text = open(MODEL_SRC).read()
    body = "\n\n".join(_member_function(text, n) for n in HOST_MODEL_FUNCTIONS)
    assert body.count(ADAM_KEY) == 6, "the Adam state key idiom changed in the reference"
    os.makedirs(GEN, exist_ok=True)
    with open(os.path.join(GEN, "ref_gaussian_model_functions.inc"), "w") as f:
        f.write(f'#line 1 "{MODEL_SRC} (extract)"\n' + body + "\n")
    base = os.path.dirname(torch.__file__)
    torch_inc = ["-I" + os.path.join(base, "include"), "-I" + os.path.join(base, "include", "torch", "csrc", "api", "include")]
    libdir = os.path.join(base, "lib")
    common = ["g++", "-std=c++17", "-O2", "-fPIC", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w"]

    def make_tree(kind):
        """the include tree of one flavour: symbolic links to the reference's files, the stand-ins of oracle/ref_host/ for the
        four headers that need Sophus / Eigen / OpenCV / ORB-SLAM3, and -- *_fused_loss -- this repository's loss_utils.h"""
        tree = os.path.join(GEN, "hosttree_" + kind)
        shutil.rmtree(tree, ignore_errors=True)
        os.makedirs(os.path.join(tree, "include"))
        os.makedirs(os.path.join(tree, "src"))
        for f in HOST_HEADERS_REF:
            src = our_loss if (f == "loss_utils.h" and kind.endswith("fused_loss")) else os.path.join(REF, "include", f)
            os.symlink(src, os.path.join(tree, "include", f))
        for f in HOST_HEADERS_STANDIN:
            os.symlink(os.path.join(standin, f), os.path.join(tree, "include", f))
        for f in HOST_SOURCES:
            os.symlink(os.path.join(REF, "src", f), os.path.join(tree, "src", f))
        os.symlink(os.path.join(REF, "third_party"), os.path.join(tree, "third_party"))
        return tree

    try:
        for kind, out in HOST_OUT.items():
            tree = make_tree(kind)
            ops = "-DREF_HOST_OPS=" + HOST_OPS[kind]
            if kind.startswith("hip"):
                # the reference's sources with NO prefix and no definitions beyond what any ROCm LibTorch consumer sets
                flags = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", ops, "-I" + compat, "-I" + tree, "-I" + os.path.join(tree, "include")] + \
                    torch_inc + ["-I/opt/rocm/include"]
                link, lib_kind = ["-ltorch_hip", "-lc10_hip"], "hip"
            else:
                flags = ["-DREF_HOST_EMU=1", ops, "-include", os.path.join(standin, "emu_device.h"), "-I" + tree,
                         "-I" + os.path.join(tree, "include")] + torch_inc
                link, lib_kind = [], "emu"
            glue_flags = ["-include", os.path.join(compat, "optimizer_key.h")]
            odir = os.path.join(GEN, "obj_" + kind)
            os.makedirs(odir, exist_ok=True)
            procs, objs = [], []
            for f in HOST_SOURCES:
                o = os.path.join(odir, f + ".o")
                procs.append(subprocess.Popen(common + flags + ["-c", os.path.join(tree, "src", f), "-o", o]))
                objs.append(o)
            o = os.path.join(odir, "ref_host.cpp.o")
            procs.append(subprocess.Popen(common + flags + glue_flags + ["-I" + GEN, "-c", glue, "-o", o]))
            objs.append(o)
            if any(p.wait() != 0 for p in procs):
                raise RuntimeError(f"the reference's host sources did not compile ({kind})")
            link_dirs = sorted({os.path.dirname(l) for l in libs[lib_kind]})
            subprocess.check_call(["g++", "-shared", "-o", out] + objs + ["-L" + d for d in link_dirs] +
                                  ["-l" + os.path.basename(l)[3:-3] for l in libs[lib_kind]] +
                                  ["-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10"] + link +
                                  # --wrap=rand: the reference loop's std::rand() draws from the harness (oracle/ref_host.cpp)
                                  ["-Wl,--no-undefined", "-Wl,--wrap=rand", "-Wl,-rpath," + libdir] + ["-Wl,-rpath," + d for d in link_dirs])
    finally:
        shutil.rmtree(GEN, ignore_errors=True)
    return dict(HOST_OUT)


if __name__ == "__main__":
    r = build(force="--force" in sys.argv)
    print(r if r else "reference sources not available and no prebuilt library")
    print(build_loss(force="--force" in sys.argv))
    print(build_densify(force="--force" in sys.argv))
    print(build_link_consumer(force="--force" in sys.argv))
    print(build_host_tree(force="--force" in sys.argv))
