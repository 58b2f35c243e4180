"""yond_public_amd -- MI355X-native (gfx950) hot path of YOND ("You Only Need a Denoiser"):
noise-level estimation, generalized-Anscombe VST / inverse VST and the AWGN raw denoiser
forward pass, as hand-written HIP kernels behind the reference's Python plugin surface."""
__version__ = "0.1.0"
