"""Backend switch kept for drop-in compatibility with the reference
(`interpol/backend.py:1`).  The external `jitfields` package is not part of this
build; setting the flag makes the API raise instead of silently ignoring it."""
jitfields = False
