"""mvsdf_amd -- MI355X (gfx950) native hot path of MVSDF: sphere tracing over the SDF MLP, SDF gradient,
surface-light-field MLP and multi-view feature consistency as hand-written HIP kernels behind the
reference's IDRNetwork / IDRLoss Python API.  See DESIGN.md."""
__version__ = '0.1.0'
