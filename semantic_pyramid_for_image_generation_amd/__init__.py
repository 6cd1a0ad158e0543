"""MI355X-native (gfx950) hot path of "Semantic Pyramid for Image Generation".

Mirrors the reference's module layout: `generator`, `discriminator`, `classifier`,
`featuresCreatorGen`, `train_model`, `eval_model`, `train_feature_creation_gen`; the operators
underneath are hand-written HIP kernels reached through the C ABI in `include/sempyr.h`.
"""
__version__ = "0.1.0"
