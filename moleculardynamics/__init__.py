"""Namespace for the MI355X-native MolecularDynamics.jl hot path (see moleculardynamics.jl_amd)."""
