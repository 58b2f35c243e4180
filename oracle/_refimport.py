"""TEST INFRASTRUCTURE ONLY -- import the read-only reference under stub modules.

Only `oracle/gen_golden.py` (run in the build container, where /root/reference
exists) uses this file.  Nothing here travels into the product path and nothing
under `yond_public_amd/` may import it.

The reference cannot be imported as shipped: `utils/utils.py:5` imports cv2,
`:27-28` skimage, `:35-36` exifread/rawpy, `:43,46` h5py/natsort, `:52` kornia;
`archs/__init__.py:2,6` imports cv2 and torchsummary; `YOND_SIDD.py:10` bm3d.
None of them is installed (no network), so empty stub modules are injected.

The one stub with arithmetic is `cv2.blur` (called at `utils/isp_algos.py:236,240`
and `YOND_SIDD.py:69,70,97`): OpenCV is a third-party dependency that is absent
from /root/reference and from this image, so its behaviour is restated from its
documentation -- normalised k x k box mean, anchor at the window centre,
BORDER_REFLECT_101, every channel filtered independently, sums accumulated in
double and the result cast back to the input dtype.  It is implemented here with
`scipy.ndimage.uniform_filter(mode='mirror')` (a different code path from
`oracle/yond_oracle.box_blur`, which uses float64 cumulative sums) so that the two
restatements cross-check each other.  PARITY IS UNPINNED AT THE cv2 BOUNDARY: no
test or golden vector in the reference pins cv2.blur.
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("YOND_REFERENCE_ROOT", "/root/reference")


def _cv2_blur(src, ksize, dst=None, anchor=None, borderType=None):
    from scipy.ndimage import uniform_filter

    kx, ky = int(ksize[0]), int(ksize[1])          # cv2 ksize is (width, height)
    a = np.asarray(src)
    size = (ky, kx) + (1,) * (a.ndim - 2)
    out = uniform_filter(a.astype(np.float64), size=size, mode="mirror")
    return out.astype(a.dtype)


def _cv2_gaussian_kernel(ksize, sigma, ktype=None):
    """cv2.getGaussianKernel restated from its documentation: G_i = alpha * exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)),
    sum G_i = 1, as a (ksize, 1) float64 column.  (Called by the reference's SSIM, YOND_SIDD.py:684; unpinned.)"""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    g = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (g / g.sum()).reshape(-1, 1)


def _cv2_filter2d(src, ddepth, kernel, dst=None, anchor=None, delta=0, borderType=None):
    """cv2.filter2D restated: CORRELATION with the kernel, anchor at its centre, BORDER_REFLECT_101, output depth =
    input depth for ddepth = -1.  (YOND_SIDD.py:686-693 crop the border away, so only the interior matters.)"""
    from scipy.ndimage import correlate
    return correlate(np.asarray(src, np.float64), np.asarray(kernel, np.float64), mode="mirror").astype(np.asarray(src).dtype)


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    cv2 = stub("cv2", setNumThreads=lambda n: None, blur=_cv2_blur, getGaussianKernel=_cv2_gaussian_kernel,
               filter2D=_cv2_filter2d)
    cv2.__stub__ = True
    stub("rawpy")
    stub("rawpy.enhance")
    stub("exifread")
    stub("h5py")
    ns = stub("natsort")
    ns.natsort = ns
    stub("kornia")
    kf = stub("kornia.filters")
    sys.modules["kornia"].filters = kf
    sk = stub("skimage")
    skm = stub("skimage.metrics",
               peak_signal_noise_ratio=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()),
               structural_similarity=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    sk.metrics = skm
    stub("torchsummary", summary=lambda *a, **k: None)
    stub("bm3d", bm3d=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    stub("lpips")


def import_reference():
    """Returns the reference's `YOND_SIDD` module object (which star-imports utils/archs)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    install_stubs()
    import matplotlib
    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    cwd = os.getcwd()
    os.chdir("/tmp")       # the reference writes ./logs etc. relative to cwd; keep the repo clean
    try:
        import YOND_SIDD as ref  # noqa
    finally:
        os.chdir(cwd)
    return ref
