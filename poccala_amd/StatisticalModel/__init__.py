"""Drop-in mirror of the reference package StatisticalModel (hot-path classes only)."""
