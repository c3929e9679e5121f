"""Shared test infrastructure (never imported by the product)."""
