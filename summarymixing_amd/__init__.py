"""summarymixing_amd — MI355X-native SummaryMixing encoder hot path (hand-written gfx950 kernels behind a
C-ABI, libsmx.so) exposed through the module surface of the SamsungLabs/SummaryMixing overlay:

    summarymixing_amd.nnet.summary_mixing.SummaryMixing            <-> speechbrain/nnet/summary_mixing.py
    summarymixing_amd.lobes.models.VanillaNN.{VanillaNN,ParallelLinear}
    summarymixing_amd.lobes.models.transformer.Conformer.{ConformerEncoder,ConformerEncoderLayer,ConvolutionModule}
    summarymixing_amd.lobes.models.transformer.Branchformer.{BranchformerEncoder,BranchformerEncoderLayer}
    summarymixing_amd.lobes.models.transformer.TransformerASR.{TransformerASR,EncoderWrapper}
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
