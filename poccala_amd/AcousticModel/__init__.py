"""Drop-in mirror of the reference package AcousticModel (hot-path helpers only)."""
