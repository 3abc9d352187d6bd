"""Corpus tooling for the MI355X path (static-automaton builder CLI)."""
