"""The one piece of the reference's `hp` flavour that touches the row-update
path: the seeding entry point of the process-global engine (the models of
distributions/hp are standalone double-precision code with no Mixture)."""
