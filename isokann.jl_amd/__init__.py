"""isokann.jl_amd -- MI355X-native engine for the ISOKANN.jl train-step hot path.

Layout:
    csrc/     HIP kernels (gfx950) + the C-ABI engine  -> lib/libisokann_hip.so
    capi.py   ctypes binding of include/isokann_hip.h
    api.py    host-side mirror of the reference interface (Iso / run / SimulationData ...)
    dp.py     data-parallel driver over N (one process per GPU, torch.distributed)

The directory name is not a Python identifier; load it with
`__graft_entry__.load_package()` (registers it as module `isokann_jl_amd`).
"""
from . import capi  # noqa: F401
from .api import (  # noqa: F401
    AdamRegularized, DomainError, FeaturesAll, FeaturesAtoms, FeaturesCoords, FeaturesPairs, HipModel, IskError, Iso, ModelSpec,
    NesterovRegularized, OptSpec, SimulationData, TransformISA, TransformPseudoInv, TransformShiftscale, batch_grads, chicoords, chis,
    coords, densenet, featuredim, features, halfinds, identity, inputdim, isotarget, koopman, koopman_f64, outputdim, pairnet,
    pairnet_widths, propcoords, propfeatures, run, train_batch, koopman_coords, chi_kchi, validationloss,
    ValidationLossLogger, koopman_variance, lastcat, mergedata, set_data, append_data, save, load, chicoords_pullback, dchidx, data_from_trajectory, data_from_trajectories,
    WeightedSamples, set_koopman_weights, rates,
)
