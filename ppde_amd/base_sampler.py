"""Sampler interface of the path (same contract as the reference's ppde/base_sampler.py:4-33)."""
from abc import ABC, abstractmethod


class BaseSampler(ABC):
    """A sampler evolves a population of one-hot sequences under an energy function."""

    @abstractmethod
    def run(self, initial_population, num_steps, energy_function, min_pos, max_pos, oracle, log_every):
        """initial_population: Tensor [n, L, 20]; min_pos/max_pos: inclusive residue range open to mutation;
        oracle: callable Tensor[n, L, 20] -> Tensor[n] used for logging only."""
        raise NotImplementedError
