"""Host-side mirror of the slice of the reference's `data/` package the hot path touches."""
