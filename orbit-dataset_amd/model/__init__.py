"""Host-side mirror of the reference's `model/` package for the episodic few-shot hot path."""
